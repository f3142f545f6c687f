import sys, os, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import branching_case
from yalla_amd import _ffi
from yalla_amd.solution import Solution
dev = _ffi.device_lib()
for n0 in (10_000, 30_000, 100_000):
    for variant in (2, 3):
        s, _ = branching_case.setup(dev, n_0=n0, n_max=int(n0 * 1.4))
        s.set_param("prolif_rate", 0.0); s.set_param("force_variant", variant)
        s.take_step(0.2, 3); s.synchronize()
        t0 = time.perf_counter(); s.take_step(0.2, 22); s.synchronize(); el = time.perf_counter() - t0
        print("branching", n0, "variant", variant, "%.3g c-u/s %.1f us/step" % (n0 * 22 / el, el / 22 * 1e6)); s.close()
