import sys, os
ROOT=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
sys.path.insert(0,ROOT); sys.path.insert(0,os.path.join(ROOT,"tests"))
import fuzz_parity
from yalla_amd import _ffi
from conftest import build_oracle
oracle, device = _ffi.bind(build_oracle()), _ffi.device_lib()
first=int(sys.argv[1]); count=int(sys.argv[2])
for seed in range(first, first+count):
    c = fuzz_parity.draw(seed)
    print(c, flush=True)
    print(fuzz_parity.run_case(oracle, device, c), flush=True)
