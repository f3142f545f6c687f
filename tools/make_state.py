#!/usr/bin/env python3
"""Set a BASELINE configuration's system up at its stated size and save it, so that profiled
runs of bench.py see the steady state only:  make_state.py <3|4> <out.npz> [cells]
  3  examples/branching.cu's cells and functor, 100 000-cell snapshot
  4  examples/passive_growth.cu grown from 200 to >= 10^6 Po_cell cells (dynamic d_n)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from yalla_amd import _ffi, cases

config, path = int(sys.argv[1]), sys.argv[2]
lib = _ffi.device_lib()
if config == 4:
    state = cases.config4_state(lib, int(sys.argv[3]) if len(sys.argv) > 3 else 1_000_000)
else:
    state = cases.config3_state(lib, int(sys.argv[3]) if len(sys.argv) > 3 else 100_000)
cases.save_state(state, path)
print({k: v for k, v in state.items() if k not in ("X", "old_v", "type")})
