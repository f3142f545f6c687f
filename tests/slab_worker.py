"""One rank of the gloo slab tests (launched by torch.distributed.run):
    slab_worker.py <out.npz> [oracle|device|rccl] [model]
The step is sequenced in C++ (ya_slab_step); `oracle` / `device`: gloo behind the transport callbacks
(device: both ranks on GPU 0, messages staged through the host); `rccl`: one GPU per rank, the
messages through libyalla_hip.so's own RCCL communicator (needs as many GPUs as ranks)."""
import os
import sys

import numpy as np
import torch.distributed as dist

from conftest import build_oracle
from yalla_amd import _ffi
from yalla_amd import slab as slab_mod
from yalla_amd.solution import Solution


def main(out, backend="oracle", model="springs_grid"):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    comm = None
    if backend == "rccl":    # one GPU per rank: the communicator selects GPU LOCAL_RANK itself
        os.environ["MASTER_PORT"] = str(int(os.environ["MASTER_PORT"]) + 7)  # its own rendezvous port
        comm = slab_mod.NativeComm(port_offset=1)
        assert (comm.rank, comm.world) == (rank, world)
        lib = _ffi.device_lib()
        n, gs = 40000, 50
    elif backend == "device":  # both ranks on GPU 0, device buffers staged through the host
        import torch
        torch.cuda.set_device(0)
        lib = _ffi.device_lib()
        n, gs = 40000, 50
    else:
        lib = _ffi.bind(build_oracle())
        n, gs = 3000, 50
    dt = 0.003 if model == "springs_grid" else 0.002
    with Solution(model, n, gs, 1.0, lib=lib) as s:
        s.random_sphere(0.5, 3)
        X0 = s.h_X[:n].copy()
    sl = slab_mod.Slab(model, X0, rank, world, gs, lib=lib)
    if model.startswith("sorting"):
        sl.sim.set_param("n_cells", n)  # the functor splits the types at the GLOBAL id n / 2
    if comm is not None:
        sl.use(comm=comm)
    else:
        sl.use(transport=slab_mod.CallbackTransport(device_memory=backend == "device"))
    for _ in range(6):
        sl.step(dt)
    gid, X = sl.own_cells()
    parts = [None] * world
    dist.gather_object((gid, X), parts if rank == 0 else None, dst=0)
    if rank == 0:
        full = np.zeros_like(X0)
        count = 0
        for g, x in parts:
            full[g] = x
            count += len(g)
        assert count == n
        np.savez(out, X0=X0, X=full)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main(*sys.argv[1:])
