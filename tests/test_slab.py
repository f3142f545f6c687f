"""z-slab decomposition (include/slab_logic.inc, yalla_amd/slab.py): a system cut into slabs must
evolve like the undivided system.  CPU: the native step on the oracle backend, several slabs in
one process (a host thread each) and across two gloo ranks.  GPU: the same on the device, several
slabs sharing the one GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest

from yalla_amd import slab as slab_mod
from yalla_amd.solution import Solution

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def reference_run(lib, n, gs, dist, seed, dt, steps, tree=False, model="springs_grid"):
    with Solution(model, n, gs, 1.0, lib=lib) as s:
        if lib.ya_models_is_device() == 0:
            s.set_reduce_order(1 if tree else 0)
        s.random_sphere(dist, seed)
        if model.startswith("sorting"):
            s.set_param("n_cells", n)
        X0 = s.h_X[:n].copy()
        s.take_step(dt, steps)
        return X0, s.positions()


def slab_run(lib, X0, world, gs, dt, steps, device="cpu", migrate_every=1, model="springs_grid",
             force_variant=None, tail_tiles=None):
    """`world` slabs of one system in this process, a host thread per slab running the native
    step (yalla_amd.slab.run_slabs); returns the positions by global id and how many cells
    changed owner."""
    plan = slab_mod.slab_plan(X0, world, 1.0, lib)
    slabs = [slab_mod.Slab(model, X0, r, world, gs, lib=lib, plan=plan) for r in range(world)]
    if force_variant is not None:
        for s in slabs:
            s.sim.set_param("force_variant", force_variant)
    if tail_tiles is not None:   # half tiles exist under the by-plane summation order only
        for s in slabs:
            assert s.sim.set_param("sum_order", 1) == 0
            s.sim.set_param("tail_tiles", tail_tiles)
    if model.startswith("sorting"):
        for s in slabs:
            s.sim.set_param("n_cells", len(X0))  # types split at the GLOBAL id n / 2 (sorting.cu:24)
    moved = 0
    owners0 = [set(s.own_cells()[0].tolist()) for s in slabs]
    slab_mod.run_slabs(slabs, dt, steps, migrate_every, device_memory=device != "cpu")
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


def check(lib, n, world, steps, dt, device="cpu", migrate_every=1, model="springs_grid", flips=0):
    """Every cell within 1e-5 of the undivided run -- except up to `flips` cells moved by a pair that sits at
    the spring's non-zero cut-off in one run and beyond it in the other (tests/slab_explain.py shows them pair
    by pair on the oracle; here only their size is bounded: a cell's velocity is O(1), so <= steps * dt)."""
    X0, Xref = reference_run(lib, n, 50, 0.5, 3, dt, steps, model=model)
    X, moved = slab_run(lib, X0, world, 50, dt, steps, device, migrate_every, model=model)
    scale = np.abs(Xref).max()
    diff = np.abs(X - Xref).max(axis=1)
    assert (diff > 1e-5 * scale).sum() <= flips, (int((diff > 1e-5 * scale).sum()), float(diff.max()))
    assert diff.max() <= (steps * dt if flips else 1e-5 * scale)
    return moved


@pytest.mark.parametrize("world,n", [(1, 3000), (2, 3000), (3, 3000), (5, 9000)])
def test_slabs_match_undivided_system_oracle(oracle, world, n):
    check(oracle, n, world, 4, 0.002)


def check_strictly(lib, n, world, steps, dt, device="cpu", migrate_every=1):
    """fading_grid: a force that fades to zero at the cut-off and friction_on_background.  Nothing about a
    pair jumps when it crosses the cut-off, so NO cell may differ: 2e-6 of the system's extent (the centre
    of mass is summed in another association across slabs, 1e-7 per step) for every cell, over as many
    steps as it takes for hundreds of cells to change their owner."""
    X0, Xref = reference_run(lib, n, 50, 0.5, 3, dt, steps, model="fading_grid")
    X, moved = slab_run(lib, X0, world, 50, dt, steps, device, migrate_every, model="fading_grid")
    scale = np.abs(Xref).max()
    assert np.abs(Xref - X0).max() > 0.1, "the system hardly moved: the case checks nothing"
    assert np.abs(X - Xref).max() <= 2e-6 * scale, np.abs(X - Xref).max() / scale
    return moved


@pytest.mark.parametrize("world,n,steps,dt,migrate_every", [(3, 3000, 30, 0.01, 1), (5, 9000, 30, 0.02, 4),
                                                            (2, 6000, 40, 0.02, 8)])
def test_slabs_match_undivided_system_strictly_oracle(oracle, world, n, steps, dt, migrate_every):
    assert check_strictly(oracle, n, world, steps, dt, migrate_every=migrate_every) > 20


def test_slabs_of_five_float_points_oracle(oracle):
    """Po_cell (20-byte points: position + polarity) through the decomposition: messages, mirrored
    rows and updates are sized by the point type."""
    assert check(oracle, 3000, 3, 4, 0.002, model="relu_po_grid") >= 0


def test_cells_migrate_between_slabs_oracle(oracle):
    moved = check(oracle, 3000, 4, 12, 0.004)
    assert moved > 0, "test too gentle: nothing crossed a slab face"


def test_postponed_migration_oracle(oracle):
    """Migrating every 4th step only: strays stay inside the ghost margin."""
    moved = check(oracle, 3000, 3, 12, 0.002, migrate_every=4)
    assert moved >= 0


def test_slabs_thinner_than_the_ghost_layer_are_refused(oracle):
    X0, _ = reference_run(oracle, 300, 50, 0.5, 3, 0.001, 0)
    with pytest.raises(slab_mod.YallaError):
        slab_mod.slab_plan(X0, 6, 1.0, oracle)


def test_native_plan_balances_own_plus_mirrored_cells(oracle):
    """ya::slab_plan through the C ABI against its numpy restatement: the cuts balance own + 0.26 *
    mirrored cells (a middle slab of a ball mirrors two cross-sections, an end slab one small cap),
    capacities from the fullest ghost layer."""
    X0, _ = reference_run(oracle, 5000, 50, 0.5, 3, 0.001, 0)
    bounds, halo_cap, mig_cap, n_max = slab_mod.slab_plan(X0, 4, 1.0, oracle)
    assert np.array_equal(bounds, slab_mod.slab_bounds(X0[:, 2], 4))
    z = X0[:, 2]

    def costs(b):  # what the plan balances
        out = []
        for r in range(4):
            own = np.count_nonzero((z >= b[r]) & (z < b[r + 1]))
            ghosts = (np.count_nonzero((z >= b[r] - 1.25) & (z < b[r])) if r > 0 else 0) + \
                     (np.count_nonzero((z >= b[r + 1]) & (z < b[r + 1] + 1.25)) if r < 3 else 0)
            out.append(own + 0.26 * ghosts)
        return out

    quantiles = slab_mod.slab_bounds(z, 4, ghost_weight=0.0)
    assert np.array_equal(quantiles[1:-1], np.sort(z)[[1250, 2500, 3750]])
    balanced, equal_own = costs(bounds), costs(quantiles)
    assert max(balanced) < max(equal_own), "the balanced cuts are no better than the quantiles"
    assert max(balanced) - min(balanced) <= 2.0, balanced      # to a cell or two
    fullest = max(max(np.count_nonzero((z >= f - 1.25) & (z < f)), np.count_nonzero((z >= f) & (z < f + 1.25)))
                  for f in bounds[1:-1])
    assert halo_cap == int(fullest * 1.15) + 64 and mig_cap == halo_cap // 4 + 64
    assert n_max >= 5000 // 4 + 2 * halo_cap


def test_id_indexed_functor_in_slabs_oracle(oracle):
    """examples/sorting.cu's functor asks `i < n / 2`: in a slab it must be given the cells'
    GLOBAL ids (ghost rows carry them; with local ids every third pair would get the wrong
    adhesion strength, a factor 3 to 9).  Two and three slabs evolve like the undivided system
    (small dt: the dense start is violent and amplifies the reordered sums otherwise)."""
    for world in (2, 3):
        moved = check(oracle, 3000, world, 6, 0.002, model="sorting_grid")
        assert moved >= 0


def test_drift_guard_reselects_the_mirrored_cells_early_oracle(oracle):
    """The caller never asks for a migration (as a model with a too large migrate_every would): the
    drift guard measures how far cells have moved since the mirrored cells were chosen, votes through
    the all-reduce and makes every rank migrate and re-select in the same step -- the run stays within
    1e-5 of the undivided system.  (Rounds 1-3 stepped on silently with a stale selection.)"""
    n, dt, steps = 3000, 0.0005, 30
    X0, Xref = reference_run(oracle, n, 50, 0.5, 3, dt, steps)
    plan = slab_mod.slab_plan(X0, 3, 1.0, oracle)
    slabs = [slab_mod.Slab("springs_grid", X0, r, 3, 50, lib=oracle, plan=plan) for r in range(3)]
    slab_mod.run_slabs(slabs, dt, steps, migrate_every=10 ** 9)
    infos = [s.info() for s in slabs]
    assert len(set(infos)) == 1, "ranks disagree about the selections they made"
    selections, by_guard, failed = infos[0]
    assert by_guard >= 1 and selections == by_guard + 1 and failed == 0
    X = np.zeros_like(X0)
    for s in slabs:
        gid, Xr = s.own_cells()
        X[gid] = Xr
        moved, predicted = s.guard_state()
        assert moved + predicted <= 0.125
        s.close()
    assert np.abs(X - Xref).max() <= 1e-5 * np.abs(Xref).max()


def test_drift_beyond_the_bound_stops_every_rank_in_the_same_step_oracle(oracle):
    """A step so violent that cells move further than (halo - cube_size) / 2 = 1/8 cube at once: no
    vote can act in time, the bound IS broken -- and every rank leaves the same step with an error
    (-11 where a cell went too far, -12 elsewhere) instead of computing on."""
    n, dt = 3000, 0.05
    X0, _ = reference_run(oracle, n, 50, 0.5, 3, dt, 0)
    plan = slab_mod.slab_plan(X0, 3, 1.0, oracle)
    slabs = [slab_mod.Slab("springs_grid", X0, r, 3, 50, lib=oracle, plan=plan) for r in range(3)]
    taken = [0, 0, 0]
    shared = slab_mod.ThreadTransport.Shared(3, False)
    for r, s in enumerate(slabs):
        s.use(transport=slab_mod.ThreadTransport(shared, r))
    codes = [None] * 3

    def work(r):
        try:
            for _ in range(6):
                slabs[r].step(dt, migrate=False)
                taken[r] += 1
        except slab_mod.YallaError as err:
            codes[r] = err.code

    import threading
    threads = [threading.Thread(target=work, args=(r,)) for r in range(3)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert all(c in (-11, -12) for c in codes) and -11 in codes, codes
    assert len(set(taken)) == 1, f"ranks stopped in different steps: {taken}"
    # with the guard switched off the same run steps on (what rounds 1-3 did)
    for s in slabs:
        s.close()


@pytest.mark.parametrize("mode", ["point", "point_xy"])
def test_fixed_points_in_slabs_oracle(oracle, mode):
    """set_fixed(i) / set_fixed_xy(i) (solvers.cuh:197-208) in a decomposed system: the rank that owns
    cell i contributes its right-hand side to the stage's all-reduce, every rank subtracts it (xy: its
    x and y with the mean's z in the first stage, the point's value in the second, as the reference
    does).  i is a GLOBAL id; the cell changes owner during the run."""
    n, dt, steps, world = 3000, 0.002, 8, 3
    X0, _ = reference_run(oracle, n, 50, 0.5, 3, dt, 0)
    bounds = slab_mod.slab_plan(X0, world, 1.0, oracle)[0]
    # a cell just below the first cut plane: it is likely to cross it
    z = X0[:, 2]
    point = int(np.argmin(np.where(z < bounds[1], bounds[1] - z, np.inf)))
    with Solution("springs_grid", n, 50, 1.0, lib=oracle) as ref:
        ref.h_X[:n] = X0
        ref.copy_to_device()
        ref.set_fixed(point) if mode == "point" else ref.set_fixed_xy(point)
        ref.take_step(dt, steps)
        Xref = ref.positions()
    assert np.abs(Xref[point] - X0[point])[:2].max() < 1e-6, "the reference run did not hold the point"
    plan = slab_mod.slab_plan(X0, world, 1.0, oracle)
    slabs = [slab_mod.Slab("springs_grid", X0, r, world, 50, lib=oracle, plan=plan) for r in range(world)]
    for s in slabs:
        s.sim.set_fixed(point) if mode == "point" else s.sim.set_fixed_xy(point)
    slab_mod.run_slabs(slabs, dt, steps, migrate_every=2)
    X = np.zeros_like(X0)
    for s in slabs:
        gid, Xr = s.own_cells()
        X[gid] = Xr
        s.close()
    assert np.abs(X - Xref).max() <= 1e-5 * np.abs(Xref).max()


def test_a_failing_rank_takes_the_others_with_it_oracle(oracle):
    """Message capacities far too small for the ghost layers (-4 on the ranks that receive them): the
    failing ranks keep stepping memory-safely until their error vote has been all-reduced, and ALL
    ranks return from that same first step -- nobody is left waiting in a collective."""
    n, world = 3000, 3
    X0, _ = reference_run(oracle, n, 50, 0.5, 3, 0.001, 0)
    bounds, halo_cap, mig_cap, n_max = slab_mod.slab_plan(X0, world, 1.0, oracle)
    slabs = []
    for r in range(world):
        own = np.flatnonzero((X0[:, 2] >= bounds[r]) & (X0[:, 2] < bounds[r + 1])).astype(np.int32)
        sim = Solution("springs_grid", n_max, 50, 1.0, lib=oracle)
        sim.h_X[:len(own)] = X0[own]
        sim.h_n = len(own)
        sim.copy_to_device()
        import ctypes as C
        assert oracle.ya_slab_init(sim._h, float(bounds[r]), float(bounds[r + 1]), 1.25,
                                   own.ctypes.data_as(C.POINTER(C.c_int))) == 0
        assert oracle.ya_slab_setup(sim._h, r, world, 16, 16) == 0   # 16 cells where hundreds are needed
        sl = slab_mod.Slab.__new__(slab_mod.Slab)
        sl.rank, sl.world, sl.sim, sl._lib, sl._h, sl._transport, sl.n_floats = r, world, sim, oracle, sim._h, None, 3
        slabs.append(sl)
    with pytest.raises(slab_mod.YallaError) as raised:
        slab_mod.run_slabs(slabs, 0.001, 3, migrate_every=1)
    codes = raised.value.all_codes
    assert len(codes) == world and all(c in (-4, -12) for c in codes) and -4 in codes, codes
    for s in slabs:
        s.close()


def failing_for_want_of_room(lib, n=3000, world=3):
    """Slabs whose MIDDLE rank has no room for mirrored cells (n_own + 2 * halo_cap > n_max: -5)."""
    import ctypes as C
    X0, _ = reference_run(lib, n, 50, 0.5, 3, 0.001, 0)
    bounds, halo_cap, mig_cap, n_max = slab_mod.slab_plan(X0, world, 1.0, lib)
    slabs = []
    for r in range(world):
        own = np.flatnonzero((X0[:, 2] >= bounds[r]) & (X0[:, 2] < bounds[r + 1])).astype(np.int32)
        room = len(own) + 8 if r == 1 else n_max
        sim = Solution("springs_grid", room, 50, 1.0, lib=lib)
        sim.h_X[:len(own)] = X0[own]
        sim.h_n = len(own)
        sim.copy_to_device()
        assert lib.ya_slab_init(sim._h, float(bounds[r]), float(bounds[r + 1]), 1.25,
                                own.ctypes.data_as(C.POINTER(C.c_int))) == 0
        assert lib.ya_slab_setup(sim._h, r, world, halo_cap, mig_cap) == 0
        sl = slab_mod.Slab.__new__(slab_mod.Slab)
        sl.rank, sl.world, sl.sim, sl._lib, sl._h, sl._transport, sl.n_floats = r, world, sim, lib, sim._h, None, 3
        slabs.append(sl)
    return slabs


def test_a_rank_without_room_for_mirrored_cells_fails_with_the_others_oracle(oracle):
    """-5: the rank that cannot keep its mirrored cells must go on sending and receiving
    every stage's rows at the sizes both ends agreed on -- its neighbours did mirror ITS cells -- until
    the error vote has been summed: all ranks return from the same step, no message has another size
    than its receiver expects (the in-process transport asserts that), nobody waits for a peer."""
    slabs = failing_for_want_of_room(oracle)
    with pytest.raises(slab_mod.YallaError) as raised:
        slab_mod.run_slabs(slabs, 0.001, 3, migrate_every=1)
    codes = raised.value.all_codes
    assert sorted(codes) == [-12, -12, -5], codes   # (in the order the ranks returned)
    for s in slabs:
        s.close()


def run_ranks(tmp_path, name, port, *worker_args):
    out = tmp_path / name
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "slab_worker.py"), str(out), *worker_args]
    proc = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    return np.load(out)


def test_two_gloo_ranks_match_undivided_system(oracle, tmp_path):
    """One process per rank over gloo, world_size 2 (oracle backend): ya_slab_step with gloo
    behind the transport callbacks."""
    got = run_ranks(tmp_path, "slab_gloo.npz", 29611, "oracle")
    X0, Xref = reference_run(oracle, 3000, 50, 0.5, 3, 0.003, 6)
    assert np.array_equal(got["X0"], X0)
    scale = np.abs(Xref).max()
    assert np.abs(got["X"] - Xref).max() <= 1e-5 * scale


def test_two_gloo_ranks_id_indexed_functor(oracle, tmp_path):
    """sorting_grid (functor indexed by global id) through the C++-sequenced step, 2 ranks."""
    got = run_ranks(tmp_path, "slab_gloo_sorting.npz", 29615, "oracle", "sorting_grid")
    X0, Xref = reference_run(oracle, 3000, 50, 0.5, 3, 0.002, 6, model="sorting_grid")
    assert np.array_equal(got["X0"], X0)
    scale = np.abs(Xref).max()
    assert np.abs(got["X"] - Xref).max() <= 1e-5 * scale


def test_rccl_id_travels_through_the_rendezvous_store(tmp_path):
    """bench.py --gpus N brings ncclUniqueId from rank 0 to the others through the store the
    launcher serves on MASTER_PORT (NativeComm.over_store): two ranks under torch.distributed.run
    and two started by hand (rank 0 then serves the store itself)."""
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.path.join(ROOT, "tests"))
    worker = os.path.join(ROOT, "tests", "store_worker.py")
    out = str(tmp_path / "torchrun")
    proc = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr", "127.0.0.1", "--master-port", "29621", worker, out],
                          env=env, capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    assert os.path.exists(out + ".0") and os.path.exists(out + ".1")
    out = str(tmp_path / "by_hand")
    procs = [subprocess.Popen([sys.executable, worker, out],
                              env=dict(env, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r),
                                       MASTER_ADDR="127.0.0.1", MASTER_PORT="29622"))
             for r in (0, 1)]
    assert [p.wait(timeout=600) for p in procs] == [0, 0]
    assert os.path.exists(out + ".0") and os.path.exists(out + ".1")


@pytest.mark.gpu
def test_two_ranks_sharing_one_gpu_match_undivided_system(device, tmp_path):
    """The one-process-per-rank path on the device, world_size 2.  RCCL refuses two ranks on
    one GPU, so the transport is gloo with the messages staged through host memory; the
    sequencing (ya_slab_step) and all device work are what bench.py runs."""
    got = run_ranks(tmp_path, "slab_gloo_device.npz", 29612, "device")
    X0, Xref = reference_run(device, 40000, 50, 0.5, 3, 0.003, 6)
    assert np.array_equal(got["X0"], X0)
    scale = np.abs(Xref).max()
    assert np.abs(got["X"] - Xref).max() <= 1e-5 * scale


@pytest.mark.gpu
def test_two_ranks_over_rccl_match_undivided_system(device, tmp_path):
    """One GPU per rank, the messages through libyalla_hip.so's own RCCL communicator (ncclSend /
    ncclRecv on the communication stream beside the interior launch, ncclAllReduce): the path
    `bench.py --gpus N` takes.  Needs two GPUs: skipped on the one-GPU boxes of this build."""
    import torch
    if torch.cuda.device_count() < 2:
        pytest.skip("one GPU here: RCCL refuses two ranks on it")
    got = run_ranks(tmp_path, "slab_rccl.npz", 29618, "rccl")
    X0, Xref = reference_run(device, 40000, 50, 0.5, 3, 0.003, 6)
    assert np.array_equal(got["X0"], X0)
    scale = np.abs(Xref).max()
    assert np.abs(got["X"] - Xref).max() <= 1e-5 * scale


@pytest.mark.gpu
@pytest.mark.parametrize("world,n,steps,dt,migrate_every", [(6, 40000, 40, 0.02, 4), (3, 20000, 30, 0.01, 1),
                                                            (8, 150000, 24, 0.02, 8)])
def test_slabs_match_undivided_system_strictly_device(device, world, n, steps, dt, migrate_every):
    assert check_strictly(device, n, world, steps, dt, device="hip", migrate_every=migrate_every) > 20


@pytest.mark.gpu
def test_id_indexed_functor_in_slabs_device(device):
    """sorting_grid in 2 and 4 slabs on the device against the undivided system."""
    for world in (2, 4):
        moved = check(device, 40000, world, 6, 0.002, device="hip", model="sorting_grid")
        assert moved >= 0


@pytest.mark.gpu
def test_rccl_binding_on_one_gpu(device):
    """A REAL one-rank RCCL communicator through libyalla_hip.so's run-time binding (dlopen, own
    declarations of the NCCL ABI): ncclGetUniqueId, ncclCommInitRank, ncclAllReduce on device
    floats and on host doubles, a grouped ncclSend + ncclRecv (to this rank itself) -- every RCCL
    entry point the N-rank path uses, as far as one GPU can run them."""
    import ctypes as C
    lib = slab_mod._core_lib()
    vp = C.c_void_p
    lib.ya_comm_unique_id.argtypes = [vp]
    lib.ya_comm_create.argtypes = [vp, C.c_int, C.c_int, C.POINTER(vp)]
    lib.ya_comm_destroy.argtypes = [vp]
    lib.ya_comm_allreduce_sum.argtypes = [vp, vp, C.c_int, vp]
    lib.ya_comm_allreduce_host.argtypes = [vp, C.POINTER(C.c_double), C.c_int, C.c_int]
    lib.ya_comm_self_exchange.argtypes = [vp, vp, vp, C.c_size_t, vp]
    lib.ya_malloc.argtypes = [C.POINTER(vp), C.c_size_t]
    lib.ya_free.argtypes = [vp]
    lib.ya_memcpy_h2d.argtypes = [vp, vp, C.c_size_t]
    lib.ya_memcpy_d2h.argtypes = [vp, vp, C.c_size_t]
    ident = C.create_string_buffer(128)
    assert lib.ya_comm_unique_id(ident) == 0
    assert any(ident.raw), "ncclGetUniqueId left the id empty"
    comm = vp()
    assert lib.ya_comm_create(ident, 0, 1, C.byref(comm)) == 0
    n = 1 << 16
    a = np.random.default_rng(3).random(n).astype(np.float32)
    d_a, d_b = vp(), vp()
    assert lib.ya_malloc(C.byref(d_a), a.nbytes) == 0 and lib.ya_malloc(C.byref(d_b), a.nbytes) == 0
    assert lib.ya_memcpy_h2d(d_a, a.ctypes.data_as(vp), a.nbytes) == 0
    # sum over one rank = identity
    assert lib.ya_comm_allreduce_sum(comm, d_a, n, None) == 0
    back = np.empty_like(a)
    assert lib.ya_device_synchronize() == 0
    assert lib.ya_memcpy_d2h(back.ctypes.data_as(vp), d_a, a.nbytes) == 0
    assert np.array_equal(a, back)
    # a message to ourselves: the grouped send/recv of ya_comm_exchange
    assert lib.ya_comm_self_exchange(comm, d_a, d_b, a.nbytes, None) == 0
    assert lib.ya_device_synchronize() == 0
    assert lib.ya_memcpy_d2h(back.ctypes.data_as(vp), d_b, a.nbytes) == 0
    assert np.array_equal(a, back)
    vals = (C.c_double * 3)(1.5, -2.0, 7.0)
    assert lib.ya_comm_allreduce_host(comm, vals, 3, 1) == 0 and list(vals) == [1.5, -2.0, 7.0]
    lib.ya_free(d_a)
    lib.ya_free(d_b)
    assert lib.ya_comm_destroy(comm) == 0


@pytest.mark.gpu
def test_rccl_communicator_single_rank(device):
    """ya_comm_* with world_size 1 on the GPU box (RCCL needs one GPU per rank): creation from
    the environment, the host all-reduce, and ya_slab_step driven through ya_slab_use_rccl."""
    env_keep = {k: os.environ.get(k) for k in ("RANK", "WORLD_SIZE")}
    os.environ["RANK"], os.environ["WORLD_SIZE"] = "0", "1"
    try:
        # (a real one-rank RCCL communicator from an id, as NativeComm.over_store makes them)
        real = slab_mod.NativeComm.from_id(slab_mod.NativeComm.unique_id(), 0, 1)
        assert (real.rank, real.world) == (0, 1) and real.allreduce_host([3.0], take_max=True) == [3.0]
        # RCCL's own account of the communicator (ya_comm_info: what bench.py prints as "rccl")
        facts = real.info()
        assert facts["kind"] == "rccl" and (facts["ranks"], facts["rank"]) == (1, 0)
        assert facts["device"] == facts["current_device"] and facts["pci"] > 0
        assert len(facts["pci_bus_id"].split(":")) == 3
        gathered = real.gather_info()
        assert gathered == [{"rank": 0, "ranks": 1, "device": facts["device"], "pci": facts["pci"],
                             "pci_bus_id": facts["pci_bus_id"].lower()}]
        real.close()
        comm = slab_mod.NativeComm()
        assert (comm.rank, comm.world) == (0, 1)
        assert comm.allreduce_host([1.5, 2.0]) == [1.5, 2.0]
        assert comm.info()["kind"] == "none"    # world 1 without an id: no RCCL communicator behind it
        X0, Xref = reference_run(device, 20000, 50, 0.5, 3, 0.003, 4)
        sl = slab_mod.Slab("springs_grid", X0, 0, 1, 50, lib=device)
        sl.use(comm=comm)
        for _ in range(4):
            sl.step(0.003)
        gid, X = sl.own_cells()
        full = np.zeros_like(X0)
        full[gid] = X
        assert np.abs(full - Xref).max() <= 1e-5 * np.abs(Xref).max()
        sl.close()
        comm.close()
    finally:
        for k, v in env_keep.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.gpu
def test_communicator_facts_gathered_over_several_ranks(device):
    """NativeComm.gather_info with more than one rank (what bench.py prints as "rccl" and checks against --gpus): four
    loopback communicators, a host thread each -- every rank ends up with the same list, one entry per rank in
    rank order.  (RCCL itself refuses two ranks on one GPU; its one-rank form is checked in
    test_rccl_communicator_single_rank.)"""
    import threading
    comms = slab_mod.loopback_comms(4)
    got, errors = [None] * 4, []

    def work(r):
        try:
            got[r] = comms[r].gather_info()
        except Exception as err:   # noqa: BLE001 -- reported below
            errors.append(err)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    assert all(g == got[0] for g in got)
    assert [g["rank"] for g in got[0]] == [0, 1, 2, 3] and all(g["ranks"] == 4 for g in got[0])
    assert len({g["pci_bus_id"] for g in got[0]}) == 1      # one GPU here: bench.py would refuse this as a real job
    assert comms[2].info()["kind"] == "loopback"
    for c in comms:
        c.close()


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_slabs_match_undivided_system_device(device, world):
    moved = check(device, 40000, world, 6, 0.004, device="hip")
    assert moved >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["springs_grid", "sorting_grid"])
def test_slabs_with_several_lanes_per_cell(device, model):
    """grid_force_coop in a slab (ghost cells get no force, functors get global ids): the same
    three slabs stepped with force_variant 3 and with the default kernel end bit-identical."""
    X0, _ = reference_run(device, 30000, 50, 0.5, 3, 0.002, 0, model=model)
    runs = [slab_run(device, X0, 3, 50, 0.002, 6, "hip", model=model, force_variant=v)[0] for v in (2, 3)]
    assert np.array_equal(runs[0].view(np.uint32), runs[1].view(np.uint32))


@pytest.mark.gpu
def test_postponed_migration_device(device):
    """Cells that left their slab are handed over every 4th step only (what bench.py does)."""
    moved = check(device, 40000, 3, 12, 0.004, device="hip", migrate_every=4)
    assert moved >= 0


@pytest.mark.gpu
def test_eight_slabs_of_a_million_cells_native_sequencing(device):
    """north_star's decomposition at scale on one GPU: 1.2 M cells in EIGHT slabs, every slab a
    Solution<float3, Slab_grid_solver> stepped by its own host thread through the native sequencing
    (tools/slab_rehearsal.cu: Slab_grid_solver::take_step with a callback transport, migration
    every 4th step), against the undivided system after the same 8 take_steps.  Tolerance: 1e-5
    of the system's extent, with tests/fuzz_slab.py's budget for pairs within rounding of the
    cut-off (each moves two cells by 0.5 dt)."""
    import json
    exe = os.path.join(ROOT, "tools", "slab_rehearsal")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "slab_rehearsal"], check=True, capture_output=True)
    n, steps, warmup, dt = 1_200_000, 6, 2, 0.001
    proc = subprocess.run([exe, str(n), "8", str(steps), str(warmup), "4"], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    out = json.loads(proc.stdout.strip().splitlines()[-1])
    assert out["world"] == 8 and len(out["slabs"]) == 8 and out["cells_after"] == n
    assert min(s["n_own"] for s in out["slabs"]) > 0.7 * n / 8   # (cuts balance own + 0.4 mirrored cells)
    assert all(s["n_ghost"] > 0 for s in out["slabs"])
    par = out["parity"]
    assert par["take_steps"] == steps + warmup and par["cells_missing"] == 0
    assert par["cells_beyond_1e-5"] <= max(4, n // 2000), par
    assert par["max_abs_diff"] <= 2.0 * (steps + warmup) * dt, par


@pytest.mark.gpu
def test_slabs_with_a_generic_force(device):
    """The decomposed step WITH generic forces (they read d_X1, so the step goes through the plain
    predictor over all local cells, the mirrored cells' d_X1 copied into the sorted copy, and the
    plain corrector -- not the sorted-copy predictor and raw corrector of the benchmarked path):
    400 000 cells in four slabs, every cell also pulled towards the origin by a generic force
    (tools/slab_rehearsal.cu, YALLA_REHEARSAL_GENERIC=1), against the undivided system."""
    import json
    exe = os.path.join(ROOT, "tools", "slab_rehearsal")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "tools"), "slab_rehearsal"], check=True, capture_output=True)
    n, steps, warmup, dt = 400_000, 6, 2, 0.001
    proc = subprocess.run([exe, str(n), "4", str(steps), str(warmup), "4"], capture_output=True, text=True,
                          timeout=600, env=dict(os.environ, YALLA_REHEARSAL_GENERIC="1"))
    assert proc.returncode == 0, proc.stdout[-2000:] + proc.stderr[-2000:]
    out = json.loads(proc.stdout.strip().splitlines()[-1])
    assert out["world"] == 4 and out["cells_after"] == n
    par = out["parity"]
    assert par["take_steps"] == steps + warmup and par["cells_missing"] == 0
    assert par["cells_beyond_1e-5"] <= max(4, n // 2000), par
    assert par["max_abs_diff"] <= 2.0 * (steps + warmup) * dt, par


@pytest.mark.gpu
def test_slabs_in_the_fast_arithmetic_tier(device):
    """libyalla_models_fast.so (contracted multiply-adds, bare v_sqrt_f32 / v_rcp_f32) carries the
    same decomposition: three slabs against the undivided system of the same tier."""
    from yalla_amd import _ffi
    fast = _ffi.device_lib("fast")
    assert fast.ya_models_arith() == 1
    # (no pair trace in this tier: a flip at the cut-off -- one was found here once the engine summed by plane
    # group, 3e-4 on two cells -- is bounded, not explained; the exact tier's cases are explained by the oracle)
    moved = check(fast, 40000, 3, 6, 0.002, device="hip", flips=4)
    assert moved >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("model", ["springs_grid", "relu_po_grid"])
def test_half_tiles_at_the_end_of_both_launches_of_a_stage(device, model):
    """grid_force_bits' tail (the last tiles of a launch as two half-tile workgroups that meet through memory)
    in the boundary and the interior launch of a slab stage, which run side by side with exchange areas of
    their own: which tiles are split changes no bit."""
    X0, _ = reference_run(device, 40000, 50, 0.5, 3, 0.002, 0, model=model)
    runs = [slab_run(device, X0, 3, 50, 0.002, 6, "hip", 2, model=model, tail_tiles=tail)[0] for tail in (0, 16, 9, 1 << 20)]   # the last: every tile of both launches as halves
    assert np.abs(runs[0] - X0).max() > 1e-3
    for X in runs[1:]:
        assert np.array_equal(runs[0].view(np.uint32), X.view(np.uint32))


@pytest.mark.gpu
def test_slabs_of_five_float_points_device(device):
    """The same on the device: 20-byte rows in the right-hand-side messages, the sorted-copy
    predictor and the raw corrector instantiated for Po_cell."""
    assert check(device, 40000, 3, 6, 0.002, device="hip", model="relu_po_grid") >= 0


@pytest.mark.gpu
@pytest.mark.parametrize("world,model", [(2, "springs_grid"), (4, "springs_grid"), (3, "relu_po_grid")])
def test_asynchronous_path_against_peers_on_one_gpu(device, world, model):
    """The decomposed step's stream choreography with real peers: every slab's exchange of right-hand
    sides runs on its own communication stream between two events, beside the interior launch on a
    third stream, the all-reduce and the update wait for it (Slab_grid_solver::stage_exchange /
    stage_join) -- the path `bench.py --gpus N` takes over RCCL.  RCCL refuses two ranks on one GPU, so
    the peers are loopback communicators (ya_comm_create_loopback: stream-ordered copies and a kernel
    all-reduce, a host thread per slab).  Bit for bit what the blocking callback transport gives, whose
    calls end with the data in place; migration and re-selection included."""
    n, dt, steps = 60000, 0.002, 9
    X0, _ = reference_run(device, n, 50, 0.5, 3, dt, 0, model=model)
    results = []
    for asynchronous in (False, True):
        plan = slab_mod.slab_plan(X0, world, 1.0, device)
        slabs = [slab_mod.Slab(model, X0, r, world, 50, lib=device, plan=plan) for r in range(world)]
        comms = slab_mod.loopback_comms(world) if asynchronous else None
        slab_mod.run_slabs(slabs, dt, steps, migrate_every=3, device_memory=True, comms=comms)
        X = np.full_like(X0, np.nan)
        for s in slabs:
            gid, Xr = s.own_cells()
            X[gid] = Xr
            s.close()
        if comms:
            for c in comms:
                c.close()
        assert not np.isnan(X).any()
        results.append(X)
    assert np.array_equal(results[0].view(np.uint32), results[1].view(np.uint32))
