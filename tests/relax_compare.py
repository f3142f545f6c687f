#!/usr/bin/env python3
"""The relaxation sequence of ya||a's tests/test_inits.cu (relaxed_sphere, then two
relaxed_cuboids on the same Solution) for one seed, on the device engine and on the CPU
oracle: do cells drift away in both?  (Test infrastructure: uses the oracle.)"""
import ctypes, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))  # run as: python tests/relax_compare.py [seed]
import numpy as np
from yalla_amd import _ffi
from yalla_amd.solution import Solution
from conftest import build_oracle

libc = ctypes.CDLL("libc.so.6")
def unit_rand():
    return libc.rand() / (2147483647 + 1.0)

def random_sphere(s, dist, seed):
    n = s.h_n
    libc.srand(seed)
    r_max = (n / 0.64) ** (1 / 3) * dist / 2
    X = np.zeros((n, 3))
    for i in range(n):
        r = r_max * unit_rand() ** (1 / 3)
        theta = np.arccos(2.0 * unit_rand() - 1)
        phi = unit_rand() * 2 * np.pi
        X[i] = (r * np.sin(theta) * np.cos(phi), r * np.sin(theta) * np.sin(phi), r * np.cos(theta))
    s.h_X[:n] = X.astype(np.float32); s.copy_to_device()

def random_cuboid(s, dist, lo, hi, seed):
    dim = np.float32(hi) - np.float32(lo)
    n = int(float(dim) ** 3 / (4.0 / 3 * np.pi * (dist / 2) ** 3) * 0.64)
    s.h_n = n
    libc.srand(seed)
    X = np.zeros((n, 3))
    for i in range(n):
        X[i] = (lo + dim * unit_rand(), lo + dim * unit_rand(), lo + dim * unit_rand())
    s.h_X[:n] = X.astype(np.float32); s.copy_to_device()

def scale(s, f):
    n = s.get_d_n(); s.copy_to_host(); s.h_X[:n] = (s.h_X[:n].astype(np.float64) * f).astype(np.float32); s.copy_to_device()

def run(lib, seed):
    out = []
    with Solution("relu_grid", 5000, 50, 1.0, lib=lib) as s:
        s.h_n = 5000; random_sphere(s, 0.6, seed); s.take_step(0.1, 2000); scale(s, 0.8 / 0.8)
        out.append(np.abs(s.positions()).max())
        random_cuboid(s, 0.8, 0.0, 9.0 / 1.0, seed + 1000); s.take_step(0.1, 1000); scale(s, 1.0)
        out.append(np.abs(s.positions()).max())
        random_cuboid(s, 0.8, 0.0, 4.0 / 0.5, seed + 2000); s.take_step(0.1, 1000); scale(s, 0.5)
        out.append(np.abs(s.positions()).max())
        return out, s.positions()

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev, Xd = run(_ffi.device_lib(), seed)
print("device extents", dev, flush=True)
ora, Xo = run(_ffi.bind(build_oracle()), seed)
print("oracle extents", ora)
print("max |device - oracle| =", np.abs(Xd - Xo).max())
