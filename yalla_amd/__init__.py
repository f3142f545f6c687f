"""yalla_amd -- MI355X-native step path for ya||a spheroid-cell models.

The engine is C++/HIP: include/*.cuh (header API, functor-templated kernels),
yalla_amd/csrc/core.hip -> libyalla_hip.so (C ABI, include/yalla_hip.h) and
yalla_amd/csrc/models.hip -> libyalla_models.so (named models, C ABI
include/yalla_models.h).  This package is the thin Python host side used by
tests/ and bench.py: a ctypes binding and a mirror of the Solution facade.
"""
from ._ffi import device_lib, bind, DEVICE_LIB, CORE_LIB  # noqa: F401
from .solution import Solution, YallaError, models  # noqa: F401
