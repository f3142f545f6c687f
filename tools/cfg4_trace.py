import sys, os, time
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import numpy as np
import growth_case
from yalla_amd import _ffi
from yalla_amd.solution import Solution
dev=_ffi.device_lib()
target=1_000_000; n_max=int(target*1.3)
gs = 2 * (int((target / 0.64) ** (1 / 3) * 0.75 / 2 * 1.25) + 4)
seed_state,_=growth_case.setup(dev,"grid",200,400)
X200,types200=seed_state.positions(),seed_state.get_prop("type",200); seed_state.close()
with Solution("passive_growth_grid", n_max, gs, 1.0, lib=dev) as s:
    s.h_n=200; s.h_X[:200]=X200; s.copy_to_device()
    s.set_prop("type", np.concatenate([types200, np.zeros(n_max-200,np.int32)]))
    s.set_param("prolif_rate",0.03); s.set_param("seed",7)
    while s.get_d_n() < target: s.take_step(0.2,10)
    s.set_param("prolif_rate",0.0)
    s.take_step(0.2,3); s.synchronize()
    t0=time.perf_counter(); s.take_step(0.2,20); s.synchronize(); el=time.perf_counter()-t0
    print(s.get_d_n(), "%.3f ms/step"%(el/20*1e3))
