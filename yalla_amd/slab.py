"""z-slab decomposition of a Grid_solver model across GPUs.

The reference is single-GPU; this is the MI355X-native extension north_star asks
for: the system is cut into slabs along z (cube id = x + gs*y + gs^2*z, so a slab
is a contiguous key range), one slab per rank / GPU.  Per Heun stage a rank

  1. packs the cells within `halo` (= cube_size + margin) of its two faces
     (ordered, deterministic selection on the device) and exchanges them with the
     slab neighbours: point-to-point over the direct xGMI links (RCCL send/recv
     via torch.distributed), never a ring collective;
  2. appends the received ghost cells, builds the grid over own + ghosts and
     evaluates forces for its own cells only;
  3. all-reduces {sum of dX, cell count as two exact floats} (a few floats) for the centre-of-mass
     fix, and updates its own cells;

and after the second stage hands over the cells that left the slab.  The device
work is the engine's (`ya_slab_*` in include/yalla_models.h); this module only
sequences it and owns the communication.  `LocalComm` runs several slabs inside
one process (tests, single-GPU validation of the device path); `DistComm` is the
one-process-per-GPU path.
"""
import ctypes as C

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


class _Buffer:
    """Fixed-capacity message buffer.  device "cpu": numpy (oracle / gloo);
    "hip": device memory from libyalla_hip.so (no torch needed, LocalComm);
    "cuda[:i]": a torch CUDA tensor (DistComm over RCCL)."""

    def __init__(self, nbytes, device):
        self.nbytes = int(nbytes)
        self.array = self.tensor = None
        self.kind = "cpu" if device == "cpu" else ("hip" if device == "hip" else "torch")
        if self.kind == "cpu":
            self.array = np.zeros(self.nbytes, dtype=np.uint8)
            self.ptr = self.array.ctypes.data
        elif self.kind == "hip":
            p = C.c_void_p()
            assert _core_lib().ya_malloc(C.byref(p), self.nbytes) == 0
            assert _core_lib().ya_memset_async(p, 0, self.nbytes, None) == 0
            self.ptr = p.value
        else:
            import torch
            self.tensor = torch.zeros(self.nbytes, dtype=torch.uint8, device=device)
            self.ptr = self.tensor.data_ptr()

    def __del__(self):
        if getattr(self, "kind", None) == "hip" and self.ptr:
            _core_lib().ya_free(C.c_void_p(self.ptr))
            self.ptr = None

    def as_tensor(self):
        if self.tensor is None:
            import torch
            self.tensor = torch.from_numpy(self.array)
        return self.tensor

    def as_float32(self):
        """Host copy of the contents as float32 (tests, LocalComm all-reduce)."""
        if self.kind == "cpu":
            return self.array.view(np.float32).copy()
        if self.kind == "hip":
            out = np.empty(self.nbytes // 4, np.float32)
            assert _core_lib().ya_memcpy_d2h(out.ctypes.data, C.c_void_p(self.ptr), self.nbytes) == 0
            return out
        return self.tensor.view(__import__("torch").float32).cpu().numpy()

    def set_float32(self, values):
        values = np.ascontiguousarray(values, dtype=np.float32)
        if self.kind == "cpu":
            self.array.view(np.float32)[:] = values
        elif self.kind == "hip":
            assert _core_lib().ya_memcpy_h2d(C.c_void_p(self.ptr), values.ctypes.data, self.nbytes) == 0
        else:
            import torch
            self.tensor.view(torch.float32).copy_(torch.from_numpy(values))

    def copy_from(self, other):
        if self.kind == "cpu":
            self.array[:] = other.array
        elif self.kind == "hip":
            assert _core_lib().ya_memcpy_d2d_async(
                C.c_void_p(self.ptr), C.c_void_p(other.ptr), self.nbytes, None) == 0
        else:
            self.tensor.copy_(other.tensor)


# callback signatures of include/yalla_models.h (ya_slab_exchange_fn, ya_slab_allreduce_fn)
EXCHANGE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_long)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int)


class NativeComm:
    """RCCL communicator of libyalla_hip.so (ya_comm_*): one per process, on the current
    device.  The unique id comes from rank 0 over TCP (RANK / WORLD_SIZE / MASTER_ADDR /
    MASTER_PORT, as torch.distributed.run sets them): no torch.distributed needed."""

    def __init__(self, port_offset=1):
        lib = _core_lib()
        lib.ya_comm_create_from_env.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        lib.ya_comm_destroy.argtypes = [C.c_void_p]
        lib.ya_comm_rank.argtypes = [C.c_void_p]
        lib.ya_comm_world.argtypes = [C.c_void_p]
        lib.ya_comm_allreduce_host.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.c_int, C.c_int]
        self._lib = lib
        handle = C.c_void_p()
        code = lib.ya_comm_create_from_env(int(port_offset), C.byref(handle))
        if code != 0:
            raise YallaError(f"ya_comm_create_from_env failed ({code})")
        self.handle = handle
        self.rank, self.world = lib.ya_comm_rank(handle), lib.ya_comm_world(handle)

    def allreduce_host(self, values, take_max=False):
        """Sum (or max) of a few host doubles over all ranks; blocking."""
        arr = (C.c_double * len(values))(*values)
        code = self._lib.ya_comm_allreduce_host(self.handle, arr, len(values), 1 if take_max else 0)
        if code != 0:
            raise YallaError(f"ya_comm_allreduce_host failed ({code})")
        return list(arr)

    def barrier(self):
        self.allreduce_host([0.0])

    def close(self):
        if self.handle:
            self._lib.ya_comm_destroy(self.handle)
            self.handle = None


class CallbackTransport:
    """torch.distributed (gloo) behind the C++-sequenced step (ya_slab_step): the engine calls
    back with raw buffer pointers; host memory (oracle) is wrapped in place, device memory is
    staged through the host (tests: RCCL refuses two ranks on one GPU)."""

    def __init__(self, device_memory):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.device_memory = device_memory
        self.exchange_fn = EXCHANGE_FN(self._exchange)
        self.allreduce_fn = ALLREDUCE_FN(self._allreduce)

    def _host_view(self, ptr, nbytes):
        return np.ctypeslib.as_array((C.c_uint8 * nbytes).from_address(ptr))

    def _read(self, ptr, nbytes):
        if not self.device_memory:
            return self._host_view(ptr, nbytes)
        out = np.empty(nbytes, np.uint8)
        assert _core_lib().ya_memcpy_d2h(out.ctypes.data, C.c_void_p(ptr), nbytes) == 0
        return out

    def _write(self, ptr, data):
        if self.device_memory:
            assert _core_lib().ya_memcpy_h2d(C.c_void_p(ptr), data.ctypes.data, data.nbytes) == 0

    def _exchange(self, ctx, kind, send_lo, recv_lo, send_hi, recv_hi, nbytes):
        try:
            torch, dist = self.torch, self.dist
            ops, landed = [], []
            for send, recv, peer in ((send_lo, recv_lo, self.rank - 1), (send_hi, recv_hi, self.rank + 1)):
                if not send or peer < 0 or peer >= self.world:
                    continue
                out = torch.from_numpy(np.ascontiguousarray(self._read(send, nbytes)))
                into = self._host_view(recv, nbytes) if not self.device_memory else np.empty(nbytes, np.uint8)
                landed.append((recv, into))
                ops.append(dist.P2POp(dist.isend, out, peer))
                ops.append(dist.P2POp(dist.irecv, torch.from_numpy(into), peer))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for recv, into in landed:
                self._write(recv, into)
            return 0
        except Exception as err:  # must not propagate into C
            import sys
            print("slab exchange callback failed:", err, file=sys.stderr)
            return 1

    def _allreduce(self, ctx, buf, count):
        try:
            data = self._read(buf, 4 * count).view(np.float32).copy()
            t = self.torch.from_numpy(data)
            self.dist.all_reduce(t)
            if self.device_memory:
                self._write(buf, data.view(np.uint8))
            else:
                self._host_view(buf, 4 * count).view(np.float32)[:] = data
            return 0
        except Exception as err:
            import sys
            print("slab all-reduce callback failed:", err, file=sys.stderr)
            return 1


def slab_bounds(z, world):
    """Cut planes that give every rank the same number of cells: rank r owns
    z in [bounds[r], bounds[r + 1]); the outer faces are at -inf / +inf."""
    z = np.sort(np.asarray(z, dtype=np.float32))
    cuts = [z[(len(z) * r) // world] for r in range(1, world)]
    return np.array([-np.inf] + cuts + [np.inf], dtype=np.float32)


class Slab:
    """One rank's share of the system."""

    def __init__(self, model, X_all, rank, world, bounds, grid_size, cube_size=1.0,
                 halo_margin=0.25, lib=None, device="cpu", slack=1.15, python_buffers=True,
                 global_ids=True):
        X_all = np.asarray(X_all, dtype=np.float32)
        self.rank, self.world = rank, world
        self.z_lo, self.z_hi = float(bounds[rank]), float(bounds[rank + 1])
        halo = cube_size * (1.0 + halo_margin)
        if world > 2 and float(np.diff(np.asarray(bounds[1:-1], dtype=np.float64)).min()) < halo:
            raise YallaError("a slab is thinner than the ghost layer (%.3g): a cell's neighbours would "
                             "sit two slabs away; use fewer slabs for this system" % halo)
        z = X_all[:, 2]
        own = np.nonzero((z >= self.z_lo) & (z < self.z_hi))[0].astype(np.int32)
        # message capacity: the same on every rank (both ends of a message must
        # agree on its size), from the fullest ghost layer of the initial state
        faces = np.asarray(bounds[1:-1], dtype=np.float32)
        fullest = 0
        for f in faces:
            fullest = max(fullest, int(np.count_nonzero((z >= f - halo) & (z < f))),
                          int(np.count_nonzero((z >= f) & (z < f + halo))))
        self.halo_cap = int(fullest * slack) + 64
        self.mig_cap = self.halo_cap // 4 + 64
        n_max = int(len(own) * slack) + 2 * self.halo_cap + 2 * self.mig_cap + 64
        self.sim = Solution(model, n_max, grid_size, cube_size, lib=lib)
        self.n_floats = self.sim.n_floats
        self.sim.h_X[: len(own)] = X_all[own]
        self.sim.h_n = len(own)
        self.sim.copy_to_device()
        lib = self.sim.lib
        self._lib, self._h = lib, self.sim._h
        _check(lib.ya_slab_init(self._h, self.z_lo, self.z_hi, halo,
                                own.ctypes.data_as(C.POINTER(C.c_int))), "ya_slab_init")
        if not global_ids:  # functors that only compare i with j: spare the id gather per pair
            self.sim.set_param("slab_global_ids", 0)
        hb = lib.ya_slab_halo_bytes(self._h, self.halo_cap)
        mb = lib.ya_slab_migrate_bytes(self._h, self.mig_cap)
        has = (rank > 0, rank < world - 1)  # neighbour below / above
        self.n_local = len(own)
        self._transport = None
        if not python_buffers:  # the step is sequenced in C++ (setup_native_step / step_native)
            return
        self.send = {("halo", d): _Buffer(hb, device) if has[d] else None for d in (0, 1)}
        self.recv = {("halo", d): _Buffer(hb, device) if has[d] else None for d in (0, 1)}
        self.send.update({("mig", d): _Buffer(mb, device) if has[d] else None for d in (0, 1)})
        self.recv.update({("mig", d): _Buffer(mb, device) if has[d] else None for d in (0, 1)})
        self.sum = _Buffer(4 * (self.n_floats + 2), device)  # sum of dX, cell count in two exact pieces

    @staticmethod
    def _p(buf):
        return C.c_void_p(buf.ptr) if buf is not None else None

    def n_own(self):
        return self._lib.ya_slab_n_own(self._h)

    # --- the step sequenced in C++ (one process per rank): ya_slab_setup / ya_slab_step ---
    def setup_native_step(self, comm=None, transport=None):
        """`comm`: a NativeComm (RCCL on the device buffers); `transport`: a CallbackTransport."""
        _check(self._lib.ya_slab_setup(self._h, self.rank, self.world, self.halo_cap, self.mig_cap),
               "ya_slab_setup")
        if comm is not None:
            _check(self._lib.ya_slab_use_rccl(self._h, comm.handle), "ya_slab_use_rccl")
        elif transport is not None:
            self._transport = transport  # keeps the ctypes callbacks alive
            _check(self._lib.ya_slab_set_transport(
                self._h, C.cast(transport.exchange_fn, C.c_void_p), C.cast(transport.allreduce_fn, C.c_void_p),
                None), "ya_slab_set_transport")

    def step_native(self, dt, migrate=True):
        code = self._lib.ya_slab_step(self._h, float(dt), 1 if migrate else 0)
        if code != 0:
            raise YallaError(f"rank {self.rank}: ya_slab_step failed ({code}): -4 = a ghost layer or the "
                             "migrating cells outgrew their message, -5 = n_max too small, "
                             "-7 = no transport, -8 = the transport failed")

    def pack_halo(self, stage):
        for d in (0, 1):
            if self.send["halo", d] is not None:
                _check(self._lib.ya_slab_pack_halo(self._h, stage, d, self._p(self.send["halo", d]),
                                                   self.halo_cap), "ya_slab_pack_halo")

    def unpack_halo(self, stage):
        n = self._lib.ya_slab_unpack_halo(self._h, stage, self._p(self.recv["halo", 0]),
                                          self._p(self.recv["halo", 1]), self.halo_cap)
        if n < 0:
            raise YallaError(f"rank {self.rank}: halo exchange failed ({n}): "
                             "-4 = a neighbour's ghost layer outgrew halo_cap, -5 = n_max too small")
        self.n_local = n

    def stage_rhs(self, stage):
        _check(self._lib.ya_slab_stage_rhs(self._h, stage), "ya_slab_stage_rhs")

    def stage_sum(self, stage):
        _check(self._lib.ya_slab_stage_sum(self._h, stage, self._p(self.sum)), "ya_slab_stage_sum")

    def stage_update(self, stage, dt):
        _check(self._lib.ya_slab_stage_update(self._h, stage, float(dt), self._p(self.sum)),
               "ya_slab_stage_update")

    def migrate_pack(self):
        n = self._lib.ya_slab_migrate_pack(self._h, self._p(self.send["mig", 0]),
                                           self._p(self.send["mig", 1]), self.mig_cap)
        if n < 0:
            raise YallaError(f"rank {self.rank}: migrate_pack failed ({n})")

    def migrate_unpack(self):
        n = self._lib.ya_slab_migrate_unpack(self._h, self._p(self.recv["mig", 0]),
                                             self._p(self.recv["mig", 1]), self.mig_cap)
        if n < 0:
            raise YallaError(f"rank {self.rank}: migrate_unpack failed ({n}): "
                             "-4 = more cells left a neighbour than mig_cap, -5 = n_max too small")

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


class LocalComm:
    """All slabs live in this process (tests; one-GPU validation of the device path)."""

    def exchange(self, slabs, kind):
        for r, s in enumerate(slabs):
            if r > 0:
                slabs[r - 1].recv[kind, 1].copy_from(s.send[kind, 0])
            if r + 1 < len(slabs):
                slabs[r + 1].recv[kind, 0].copy_from(s.send[kind, 1])

    def allreduce(self, slabs):
        parts = [s.sum.as_float32() for s in slabs]
        total = parts[0].copy()
        for p in parts[1:]:
            total = total + p
        for s in slabs:
            s.sum.set_float32(total)


class DistComm:
    """One slab per process: RCCL ("nccl") on GPUs, gloo on CPU (oracle tests).  With
    gloo and device buffers (two test ranks sharing one GPU: RCCL refuses that) the
    messages are staged through host memory."""

    def __init__(self):
        import torch.distributed as dist
        self.dist = dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.stage_through_host = dist.get_backend() == "gloo"

    def exchange(self, slabs, kind):
        (s,) = slabs
        dist = self.dist
        ops, staged = [], []
        for d, peer in ((0, self.rank - 1), (1, self.rank + 1)):
            if s.send[kind, d] is None:
                continue
            out, into = s.send[kind, d].as_tensor(), s.recv[kind, d].as_tensor()
            if self.stage_through_host and out.is_cuda:
                host_in = out.cpu().clone()
                staged.append((into, host_in))
                out, into = out.cpu(), host_in
            ops.append(dist.P2POp(dist.isend, out, peer))
            ops.append(dist.P2POp(dist.irecv, into, peer))
        if ops:
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        for device_tensor, host_in in staged:
            device_tensor.copy_(host_in)

    def allreduce(self, slabs):
        import torch
        (s,) = slabs
        t = s.sum.as_tensor().view(torch.float32)
        if self.stage_through_host and t.is_cuda:
            host = t.cpu()
            self.dist.all_reduce(host)
            t.copy_(host)
        else:
            self.dist.all_reduce(t)


def step(slabs, comm, dt, migrate=True):
    """One take_step of the decomposed system (both Heun stages, then migration).
    `migrate=False` postpones the hand-over of cells that left their slab: legal
    while no cell has strayed further than the halo margin (0.25 cube_size by
    default) beyond its slab since the last migration -- the ghost layer covers
    it; callers that skip must migrate every few steps."""
    for stage in (1, 2):
        for s in slabs:
            s.pack_halo(stage)
        comm.exchange(slabs, "halo")
        for s in slabs:
            s.unpack_halo(stage)
            s.stage_rhs(stage)
            s.stage_sum(stage)
        comm.allreduce(slabs)
        for s in slabs:
            s.stage_update(stage, dt)
    if not migrate:
        return
    for s in slabs:
        s.migrate_pack()
    comm.exchange(slabs, "mig")
    for s in slabs:
        s.migrate_unpack()
