"""One rank of the gloo slab tests (launched by torch.distributed.run):
    slab_worker.py <out.npz> [oracle|device] [python|native] [model]
`python`: the step sequenced by yalla_amd/slab.py (phase by phase through the C ABI);
`native`: the step sequenced in C++ (ya_slab_step) with gloo behind the transport callbacks."""
import os
import sys

import numpy as np
import torch.distributed as dist

from conftest import build_oracle
from yalla_amd import _ffi
from yalla_amd import slab as slab_mod
from yalla_amd.solution import Solution


def main(out, backend="oracle", sequencing="python", model="springs_grid"):
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    if backend == "device":  # both ranks on GPU 0, device buffers staged through the host
        import torch
        torch.cuda.set_device(0)
        lib = _ffi.device_lib()
        n, gs, buffers = 40000, 50, "cuda:0"
    else:
        lib = _ffi.bind(build_oracle())
        n, gs, buffers = 3000, 50, "cpu"
    dt = 0.003 if model == "springs_grid" else 0.002
    with Solution(model, n, gs, 1.0, lib=lib) as s:
        s.random_sphere(0.5, 3)
        X0 = s.h_X[:n].copy()
    bounds = slab_mod.slab_bounds(X0[:, 2], world)
    native = sequencing == "native"
    sl = slab_mod.Slab(model, X0, rank, world, bounds, gs, lib=lib, device=buffers,
                       python_buffers=not native)
    if model.startswith("sorting"):
        sl.sim.set_param("n_cells", n)  # the functor splits the types at the GLOBAL id n / 2
    if native:
        sl.setup_native_step(transport=slab_mod.CallbackTransport(device_memory=backend == "device"))
        for _ in range(6):
            sl.step_native(dt)
    else:
        comm = slab_mod.DistComm()
        for _ in range(6):
            slab_mod.step([sl], comm, dt)
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
