"""z-slab decomposition of a Grid_solver model across GPUs: the host side.

The reference is single-GPU; this is the MI355X-native extension north_star asks for.  The
system is cut into slabs along z (cube id = x + gs*y + gs^2*z, so a slab is a contiguous key
range), one slab per rank / GPU.  Everything on the data path is native: the cut planes and
capacities (ya::slab_plan), the device work and its sequencing (ya_slab_step in
include/yalla_models.h: mirrored ghost cells, one message of right-hand sides per neighbour and
Heun stage travelling beside the interior tiles' forces, the all-reduce for the centre-of-mass
fix, migration), the transport (RCCL send/recv point-to-point over the direct xGMI links,
include/yalla_hip.h ya_comm_*).  This module only hands a rank its cells and picks the transport:

  NativeComm          one process per GPU, RCCL through libyalla_hip.so (bench.py --gpus N)
  CallbackTransport   one process per rank over torch.distributed / gloo (CPU tests; two test
                      ranks sharing one GPU, messages staged through the host)
  run_slabs           several slabs inside ONE process, a host thread per slab (tests; one-GPU
                      validation of the device path)
"""
import ctypes as C
import threading

import numpy as np

from .solution import Solution, YallaError, _check


_core = None


def _core_lib():
    """libyalla_hip.so through ctypes (device buffers without torch)."""
    global _core
    if _core is None:
        from . import _ffi
        _core = C.CDLL(_ffi.CORE_LIB, mode=C.RTLD_LOCAL)
        _core.ya_malloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t]
        _core.ya_free.argtypes = [C.c_void_p]
        _core.ya_memset_async.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p]
        _core.ya_memcpy_d2d_async.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
        _core.ya_memcpy_d2h.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        _core.ya_memcpy_h2d.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
    return _core


# callback signatures of include/yalla_models.h (ya_slab_exchange_fn, ya_slab_allreduce_fn)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_long,
                          C.c_void_p, C.c_long, C.c_void_p, C.c_long)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)


COMM_ID_BYTES = 128   # YA_COMM_ID_BYTES (include/yalla_hip.h) = sizeof(ncclUniqueId)


class NativeComm:
    """RCCL communicator of libyalla_hip.so (ya_comm_*): one per process.  With WORLD_SIZE > 1 it
    first puts the process on GPU LOCAL_RANK (one process per GPU), so create it BEFORE any
    Solution.  The unique id comes from rank 0 over TCP (RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT, as torch.distributed.run sets them): no torch.distributed needed.
    `NativeComm.over_store()` hands the id over through the rendezvous store the launcher
    already runs on MASTER_PORT instead (no second port)."""

    def __init__(self, port_offset=1, _handle=None):
        lib = self._bind()
        self._lib = lib
        if _handle is None:
            _handle = C.c_void_p()
            code = lib.ya_comm_create_from_env(int(port_offset), C.byref(_handle))
            if code != 0:
                raise YallaError(f"ya_comm_create_from_env failed ({code})")
        self.handle = _handle
        self.rank, self.world = lib.ya_comm_rank(_handle), lib.ya_comm_world(_handle)

    @staticmethod
    def _bind():
        lib = _core_lib()
        lib.ya_comm_create_from_env.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.ya_comm_unique_id.argtypes = [C.c_void_p]
        lib.ya_comm_create.argtypes = [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
        lib.ya_comm_destroy.argtypes = [C.c_void_p]
        lib.ya_comm_rank.argtypes = [C.c_void_p]
        lib.ya_comm_world.argtypes = [C.c_void_p]
        lib.ya_comm_allreduce_host.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]
        lib.ya_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.c_char_p]
        return lib

    @classmethod
    def from_id(cls, ident, rank, world):
        """The communicator of `world` ranks from a unique id (NativeComm.unique_id() of rank 0,
        brought here by whatever the program has: a file, MPI, a rendezvous store).  The caller
        has put the process on its GPU."""
        lib = cls._bind()
        handle = C.c_void_p()
        buf = C.create_string_buffer(bytes(ident), COMM_ID_BYTES)
        code = lib.ya_comm_create(buf, int(rank), int(world), C.byref(handle))
        if code != 0:
            raise YallaError(f"ya_comm_create failed ({code})")
        return cls(_handle=handle)

    @classmethod
    def unique_id(cls):
        buf = C.create_string_buffer(COMM_ID_BYTES)
        code = cls._bind().ya_comm_unique_id(buf)
        if code != 0:
            raise YallaError(f"ya_comm_unique_id failed ({code})")
        return buf.raw

    @staticmethod
    def _id_over_store(rank, world, key, make_id):
        import torch.distributed as dist
        from datetime import timedelta
        # (a rank may be minutes behind the others on a fresh box: the first import of the runtime)
        store, _, _ = next(iter(dist.rendezvous("env://", rank=rank, world_size=world,
                                                timeout=timedelta(minutes=30))))
        if rank == 0:
            store.set(key, make_id())
        return bytes(store.get(key)), store   # get() blocks until rank 0 has set the key

    @classmethod
    def over_store(cls, rank, world, key="yalla_rccl_id"):
        """One communicator per process, the id handed over through torch.distributed's
        rendezvous store (env://: the store torch.distributed.run serves on MASTER_PORT, or one
        rank 0 opens there) -- no process group, no second port.  The process must already be
        on its GPU (torch.cuda.set_device(LOCAL_RANK))."""
        if world <= 1:
            return cls()
        ident, store = cls._id_over_store(rank, world, key, cls.unique_id)
        comm = cls.from_id(ident, rank, world)
        comm._store = store      # rank 0 may be the store's server: keep it alive with the communicator
        return comm

    def allreduce_host(self, values, take_max=False):
        """Sum (or max) of a few host doubles over all ranks; blocking."""
        arr = (C.c_double * len(values))(*values)
        code = self._lib.ya_comm_allreduce_host(self.handle, arr, len(values), 1 if take_max else 0)
        if code != 0:
            raise YallaError(f"ya_comm_allreduce_host failed ({code})")
        return list(arr)

    def barrier(self):
        self.allreduce_host([0.0])

    def info(self):
        """What RCCL itself says about this communicator (ya_comm_info: ncclCommCount, ncclCommUserRank,
        ncclCommCuDevice) and where that device sits on the PCI bus."""
        raw = (C.c_int * 8)()
        bus = C.create_string_buffer(32)
        code = self._lib.ya_comm_info(self.handle, raw, bus)
        if code != 0:
            raise YallaError(f"ya_comm_info failed ({code})")
        return {"ranks": raw[0], "rank": raw[1], "device": raw[2], "current_device": raw[3],
                "kind": {0: "none", 1: "rccl", 2: "loopback"}[raw[4]], "pci": raw[5], "pci_bus_id": bus.value.decode()}

    def gather_info(self):
        """info() of every rank, gathered over the communicator itself (each rank fills its own slots
        of a host all-reduce): [{rank, ranks, device, pci, pci_bus_id}, ...] in rank order, the same on
        every rank.  Raises if the ranks disagree about the communicator's size."""
        mine = self.info()
        assert self.world * 4 <= 64, "ya_comm_allreduce_host carries 64 values"
        slots = [0.0] * (4 * self.world)
        slots[4 * self.rank: 4 * self.rank + 4] = [float(mine["ranks"]), float(mine["rank"]),
                                                   float(mine["device"]), float(mine["pci"])]
        slots = self.allreduce_host(slots)
        out = []
        for r in range(self.world):
            ranks, rank, device, pci = (int(v) for v in slots[4 * r: 4 * r + 4])
            out.append({"rank": rank, "ranks": ranks, "device": device, "pci": pci,
                        "pci_bus_id": "%04x:%02x:%02x.%x" % (pci >> 16, (pci >> 8) & 0xff, (pci >> 3) & 0x1f, pci & 7)})
        return out

    def close(self):
        if self.handle:
            self._lib.ya_comm_destroy(self.handle)
            self.handle = None


def loopback_comms(world):
    """`world` NativeComm handles for the slabs of THIS process on one GPU (ya_comm_create_loopback):
    stream-ordered device-to-device messages and a kernel all-reduce behind the RCCL entry points, so that
    the decomposed step's asynchronous path (the exchange on its own stream beside the interior launch)
    runs against real peers where RCCL cannot (it refuses two ranks on one GPU).  One host thread per
    handle (run_slabs(..., comms=...)); close() every one."""
    lib = NativeComm._bind()
    lib.ya_comm_create_loopback.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    handles = (C.c_void_p * world)()
    code = lib.ya_comm_create_loopback(int(world), handles)
    if code != 0:
        raise YallaError(f"ya_comm_create_loopback failed ({code})")
    return [NativeComm(_handle=C.c_void_p(h)) for h in handles]


class _Memory:
    """Reading / writing the raw buffers the engine hands to a transport callback: host memory
    on the oracle, device memory on the HIP engine (blocking copies through libyalla_hip.so)."""

    def __init__(self, device_memory):
        self.device_memory = device_memory

    def read(self, ptr, nbytes):
        if nbytes <= 0:
            return np.empty(0, np.uint8)
        if not self.device_memory:
            return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr)).copy()
        out = np.empty(nbytes, np.uint8)
        assert _core_lib().ya_memcpy_d2h(out.ctypes.data, C.c_void_p(ptr), nbytes) == 0
        return out

    def write(self, ptr, data):
        if data.nbytes == 0:
            return
        data = np.ascontiguousarray(data)
        if self.device_memory:
            assert _core_lib().ya_memcpy_h2d(C.c_void_p(ptr), data.ctypes.data, data.nbytes) == 0
        else:
            C.memmove(ptr, data.ctypes.data, data.nbytes)


class CallbackTransport:
    """torch.distributed (gloo) behind the C++-sequenced step (ya_slab_step): the engine calls
    back with raw buffer pointers and sizes; device memory is staged through the host (tests:
    RCCL refuses two ranks on one GPU)."""

    def __init__(self, device_memory):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.mem = _Memory(device_memory)
        self.exchange_fn = EXCHANGE_FN(self._exchange)
        self.allreduce_fn = ALLREDUCE_FN(self._allreduce)

    def _exchange(self, ctx, kind, send_lo, send_lo_bytes, recv_lo, recv_lo_bytes, send_hi, send_hi_bytes,
                  recv_hi, recv_hi_bytes):
        try:
            torch, dist = self.torch, self.dist
            ops, landed = [], []
            for send, n_out, recv, n_in, peer in ((send_lo, send_lo_bytes, recv_lo, recv_lo_bytes, self.rank - 1),
                                                  (send_hi, send_hi_bytes, recv_hi, recv_hi_bytes, self.rank + 1)):
                if peer < 0 or peer >= self.world:
                    continue
                if n_out > 0:
                    ops.append(dist.P2POp(dist.isend, torch.from_numpy(self.mem.read(send, n_out)), peer))
                if n_in > 0:
                    into = np.empty(n_in, np.uint8)
                    landed.append((recv, into))
                    ops.append(dist.P2POp(dist.irecv, torch.from_numpy(into), peer))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for recv, into in landed:
                self.mem.write(recv, into)
            return 0
        except Exception as err:  # must not propagate into C
            import sys
            print("slab exchange callback failed:", err, file=sys.stderr)
            return 1

    def _allreduce(self, ctx, buf, count):
        try:
            data = self.mem.read(buf, 4 * count).view(np.float32).copy()
            t = self.torch.from_numpy(data)
            self.dist.all_reduce(t)
            self.mem.write(buf, data.view(np.uint8))
            return 0
        except Exception as err:
            import sys
            print("slab all-reduce callback failed:", err, file=sys.stderr)
            return 1


class ThreadTransport:
    """The transport of `world` slabs that live in ONE process, a host thread each (run_slabs):
    a message is a copy between the slabs' buffers at a barrier, the all-reduce a host sum."""

    class Shared:
        def __init__(self, world, device_memory):
            self.world = world
            self.barrier = threading.Barrier(world)
            self.mem = _Memory(device_memory)
            self.out = [None] * world   # per rank: (to lower, to upper) message bytes
            self.sums = [None] * world
            self.failed = False

    def __init__(self, shared, rank):
        self.shared, self.rank = shared, rank
        self.exchange_fn = EXCHANGE_FN(self._exchange)
        self.allreduce_fn = ALLREDUCE_FN(self._allreduce)

    def _wait(self):
        try:
            self.shared.barrier.wait(timeout=600)
        except threading.BrokenBarrierError:
            self.shared.failed = True
            raise

    def _exchange(self, ctx, kind, send_lo, send_lo_bytes, recv_lo, recv_lo_bytes, send_hi, send_hi_bytes,
                  recv_hi, recv_hi_bytes):
        try:
            sh = self.shared
            sh.out[self.rank] = (sh.mem.read(send_lo, send_lo_bytes), sh.mem.read(send_hi, send_hi_bytes))
            self._wait()
            if self.rank > 0 and recv_lo_bytes > 0:
                data = sh.out[self.rank - 1][1]
                assert data.nbytes == recv_lo_bytes, "lower neighbour sent another size than expected"
                sh.mem.write(recv_lo, data)
            if self.rank + 1 < sh.world and recv_hi_bytes > 0:
                data = sh.out[self.rank + 1][0]
                assert data.nbytes == recv_hi_bytes, "upper neighbour sent another size than expected"
                sh.mem.write(recv_hi, data)
            self._wait()
            return 0
        except Exception as err:
            import sys
            print(f"slab exchange (rank {self.rank}) failed:", repr(err), file=sys.stderr)
            self.shared.barrier.abort()
            return 1

    def _allreduce(self, ctx, buf, count):
        try:
            sh = self.shared
            sh.sums[self.rank] = sh.mem.read(buf, 4 * count).view(np.float32).copy()
            self._wait()
            total = sh.sums[0].copy()
            for part in sh.sums[1:]:
                total = total + part   # the same order on every rank
            self._wait()
            sh.mem.write(buf, total.view(np.uint8))
            return 0
        except Exception as err:
            import sys
            print(f"slab all-reduce (rank {self.rank}) failed:", repr(err), file=sys.stderr)
            self.shared.barrier.abort()
            return 1


def slab_plan(X, world, cube_size=1.0, lib=None):
    """ya::slab_plan through the C ABI: (bounds[world + 1], halo_cap, mig_cap, n_max).  Rank r owns
    z in [bounds[r], bounds[r + 1]); the outer faces are at -inf / +inf."""
    from . import _ffi
    lib = lib if lib is not None else _ffi.device_lib()
    X = np.ascontiguousarray(X, dtype=np.float32)
    bounds = np.zeros(world + 1, np.float32)
    caps = np.zeros(4, np.int32)
    code = lib.ya_slab_plan(X.ctypes.data_as(C.POINTER(C.c_float)), X.shape[1], X.shape[0], int(world),
                            float(cube_size), bounds.ctypes.data_as(C.POINTER(C.c_float)),
                            caps.ctypes.data_as(C.POINTER(C.c_int)))
    if code == -9:
        raise YallaError("a slab is thinner than the ghost layer: a cell's neighbours would sit two slabs "
                         "away; use fewer slabs for this system")
    _check(code, "ya_slab_plan")
    return bounds, int(caps[0]), int(caps[1]), int(caps[2])


def slab_bounds(z, world, cube_size=1.0, margin=0.25, ghost_weight=0.26):
    """The cut planes alone: numpy restatement of ya::slab_plan_sorted (include/slab_logic.inc) for
    tests.  Cuts at cells' z such that the largest own + ghost_weight * mirrored cell count of a slab
    is smallest (greedy walk under a bisected bound); ghost_weight 0: the z-quantiles of rounds 1-3."""
    f32 = np.float32
    z = np.sort(np.asarray(z, dtype=f32))
    n = len(z)
    halo = f32(cube_size) * (f32(1.0) + f32(margin))
    w = float(f32(ghost_weight))

    def below(v):
        return int(np.searchsorted(z, f32(v), side="left"))

    quantiles = [(n * r) // world for r in range(world + 1)]
    best = quantiles
    if w > 0 and world > 1 and n >= world:
        def cost(a, b):
            ghosts = 0
            if a > 0:
                ghosts += a - below(z[a] - halo)
            if b < n:
                ghosts += below(z[b] + halo) - b
            return float(b - a) + w * ghosts

        def fits(T):
            cut = [0]
            for r in range(world - 1):
                a = cut[r]
                lo, hi = a + 1, n - (world - 1 - r)
                if lo > hi or cost(a, lo) > T:
                    return None
                while lo < hi:
                    mid = lo + (hi - lo + 1) // 2
                    if cost(a, mid) <= T:
                        lo = mid
                    else:
                        hi = mid - 1
                cut.append(lo)
            cut.append(n)
            return cut if cost(cut[world - 1], n) <= T else None

        lo, hi = 0.0, float(n) * (1.0 + 2.0 * w) + 1.0
        cut = fits(hi)
        if cut is not None:
            best = cut
            for _ in range(50):
                mid = 0.5 * (lo + hi)
                cut = fits(mid)
                if cut is not None:
                    hi, best = mid, cut
                else:
                    lo = mid
    return np.array([-np.inf] + [z[min(best[r], n - 1)] for r in range(1, world)] + [np.inf], dtype=f32)


class Slab:
    """One rank's share of the system: its own cells with their global ids in a Solution of the
    model, decomposed (ya_slab_init / ya_slab_setup) and given a transport."""

    def __init__(self, model, X_all, rank, world, grid_size, cube_size=1.0, lib=None, global_ids=True, plan=None):
        X_all = np.ascontiguousarray(X_all, dtype=np.float32)
        self.rank, self.world = rank, world
        bounds, self.halo_cap, self.mig_cap, n_max = plan if plan is not None else slab_plan(X_all, world, cube_size, lib)
        self.z_lo, self.z_hi = float(bounds[rank]), float(bounds[rank + 1])
        self.sim = Solution(model, n_max, grid_size, cube_size, lib=lib)
        self.n_floats = self.sim.n_floats
        assert X_all.shape[1] == self.n_floats, "the whole system's points must have the model's point type"
        lib = self.sim.lib
        self._lib, self._h = lib, self.sim._h
        # this rank's cells, their global ids, ya_slab_init and ya_slab_setup: all native
        code = lib.ya_slab_decompose(self._h, X_all.ctypes.data_as(C.POINTER(C.c_float)), len(X_all), rank, world,
                                     float(cube_size))
        if code == -9:
            raise YallaError("a slab is thinner than the ghost layer: use fewer slabs for this system")
        _check(code, "ya_slab_decompose")
        if not global_ids:  # functors that only compare i with j: spare the id gather per pair
            self.sim.set_param("slab_global_ids", 0)
        self._transport = None

    def use(self, comm=None, transport=None):
        """`comm`: a NativeComm (RCCL on the device buffers); `transport`: an object with
        exchange_fn / allreduce_fn (CallbackTransport, ThreadTransport)."""
        if comm is not None:
            _check(self._lib.ya_slab_use_rccl(self._h, comm.handle), "ya_slab_use_rccl")
        elif transport is not None:
            self._transport = transport  # keeps the ctypes callbacks alive
            _check(self._lib.ya_slab_set_transport(
                self._h, C.cast(transport.exchange_fn, C.c_void_p), C.cast(transport.allreduce_fn, C.c_void_p),
                None), "ya_slab_set_transport")

    def n_own(self):
        return self._lib.ya_slab_n_own(self._h)

    def n_local(self):
        return self._lib.ya_slab_n_local(self._h)

    def step(self, dt, migrate=True):
        code = self._lib.ya_slab_step(self._h, float(dt), 1 if migrate else 0)
        if code != 0:
            err = YallaError(f"rank {self.rank}: ya_slab_step failed ({code}): -4 = a ghost layer or the "
                             "migrating cells outgrew their message, -5 = n_max too small, -6 = a cell left "
                             "through an outer face, -7 = no transport, -8 = the transport failed, -11 = a cell "
                             "of this rank drifted further than (halo - cube_size) / 2 between two selections of "
                             "the mirrored cells, -12 = another rank failed")
            err.code = code
            raise err

    def info(self):
        """(selections of the mirrored cells so far, those the drift guard asked for, sticky failure code)"""
        return tuple(int(self._lib.ya_slab_info(self._h, k)) for k in range(3))

    def guard_state(self):
        """The drift guard's (moved, predicted) of the last step: the largest |z - z at selection| the step
        began with and the largest predictor |dz|, over this rank's own and mirrored cells."""
        return tuple(self._lib.ya_slab_info(self._h, k) * 1e-6 for k in (3, 4))

    def own_cells(self):
        """(global ids, positions) of the cells this rank owns now."""
        n = self.n_own()
        X = np.empty((max(n, 1), self.n_floats), np.float32)
        gid = np.empty(max(n, 1), np.int32)
        got = self._lib.ya_slab_get_own(self._h, X.ctypes.data_as(C.POINTER(C.c_float)),
                                        gid.ctypes.data_as(C.POINTER(C.c_int)))
        return gid[:got].copy(), X[:got].copy()

    def close(self):
        self.sim.close()


def run_slabs(slabs, dt, steps, migrate_every=1, device_memory=False, comms=None):
    """`steps` take_steps of all slabs of one system inside this process: a host thread per slab
    runs the native step (ya_slab_step) with a ThreadTransport -- or, with `comms` (loopback_comms), with
    the communicators' stream-ordered transport, i.e. through the step's asynchronous path.  Migration
    every `migrate_every`-th step and after the last one."""
    world = len(slabs)
    shared = ThreadTransport.Shared(world, device_memory)
    for r, s in enumerate(slabs):
        if comms is not None:
            s.use(comm=comms[r])
        else:
            s.use(transport=ThreadTransport(shared, r))
    errors = []

    def work(s):
        try:
            for k in range(steps):
                s.step(dt, migrate=(k + 1) % migrate_every == 0 or k == steps - 1)
        except Exception as err:   # a failed rank must not leave the others at a barrier
            errors.append(err)
            # (failures the ranks share through the all-reduce end every rank's step at the same point:
            # give the others the moment they need to get there before the barrier is broken)
            import time
            time.sleep(0.5 if getattr(err, "code", 0) in (-4, -5, -6, -11, -12) else 0.0)
            shared.barrier.abort()

    threads = [threading.Thread(target=work, args=(s,)) for s in slabs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    if errors:
        first = errors[0]
        first.all_codes = [getattr(e, "code", None) for e in errors]
        raise first
