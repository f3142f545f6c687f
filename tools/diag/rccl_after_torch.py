"""Does a one-rank RCCL communicator through libyalla_hip.so come up after torch has been imported
and used in the same process (what bench.py --gpus N does)?  rccl_after_torch.py <mode>
modes: none | import | count | use | yalla_first (the engine loaded and used BEFORE torch)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
mode = sys.argv[1]
if mode == "yalla_first":
    from yalla_amd.solution import Solution
    with Solution("springs_grid", 1000, 20, 1.0) as sim:
        sim.random_sphere(0.5, 1)
        sim.take_step(0.001, 2)
    import torch
    print("devices", torch.cuda.device_count(), "sum", float(torch.ones(8, device="cuda").sum()), flush=True)
elif mode != "none":
    import torch
    if mode in ("count", "use"):
        print("devices", torch.cuda.device_count(), flush=True)
    if mode == "use":
        torch.cuda.set_device(0)
        print("sum", float(torch.ones(8, device="cuda").sum()), flush=True)
from yalla_amd import slab

try:
    comm = slab.NativeComm.from_id(slab.NativeComm.unique_id(), 0, 1)
    print(mode, "communicator ok", comm.allreduce_host([2.5]), flush=True)
    comm.close()
except Exception as err:   # noqa: BLE001
    print(mode, "FAILED", err, flush=True)
libs = sorted({l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l or "amdhip" in l or "hsa-runtime" in l})
print(mode, libs, flush=True)
