"""Host-side mirror of ya||a's `Solution<Pt, Solver>` facade (reference
include/solvers.cuh:56-106) over the model-harness C ABI.

Names and meanings follow the reference: `h_X` is the host mirror, `h_n` the
host-side point count, `copy_to_device()` / `copy_to_host()` move n_max points
and n, `take_step(dt)` advances by one two-stage Heun step, `set_fixed*`
select what is held fixed, `get_d_n()` reads the device-side count.  A model
name selects the point type, the solver and the pairwise functor (C++ template
arguments in the reference), e.g. "springs_grid" = Solution<float3,
Grid_solver> stepping `spring`.
"""
import ctypes as C

import numpy as np

from . import _ffi


class YallaError(RuntimeError):
    pass


def _check(code, what):
    if code != 0:
        raise YallaError(f"{what} failed with harness code {code}")


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int))


class Solution:
    def __init__(self, model, n_max, grid_size=50, cube_size=1.0, lib=None):
        self.lib = lib if lib is not None else _ffi.device_lib()
        self.model = model
        self.grid_size = int(grid_size)
        handle = C.c_void_p()
        code = self.lib.ya_sim_create(model.encode(), int(n_max), int(grid_size),
                                      float(cube_size), C.byref(handle))
        if code == -1:
            raise YallaError(f"unknown model {model!r}; known: {models(self.lib)}")
        _check(code, "ya_sim_create")
        self._h = handle
        self.n_max = self.lib.ya_sim_n_max(self._h)
        self.n_floats = self.lib.ya_sim_n_floats(self._h)
        ptr = self.lib.ya_sim_h_X(self._h)
        self.h_X = np.ctypeslib.as_array(ptr, shape=(self.n_max, self.n_floats))

    def close(self):
        if getattr(self, "_h", None):
            self.h_X = None
            self.lib.ya_sim_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # --- Solution facade -------------------------------------------------
    @property
    def h_n(self):
        return self.lib.ya_sim_get_h_n(self._h)

    @h_n.setter
    def h_n(self, n):
        _check(self.lib.ya_sim_set_h_n(self._h, int(n)), "set h_n")

    def copy_to_device(self):
        _check(self.lib.ya_sim_copy_to_device(self._h), "copy_to_device")

    def copy_to_host(self):
        _check(self.lib.ya_sim_copy_to_host(self._h), "copy_to_host")

    def get_d_n(self):
        return self.lib.ya_sim_get_d_n(self._h)

    def take_step(self, dt, steps=1):
        _check(self.lib.ya_sim_take_steps(self._h, float(dt), int(steps)), "take_step")

    def synchronize(self):
        _check(self.lib.ya_sim_synchronize(self._h), "synchronize")

    def set_fixed(self, point_id=None):
        if point_id is None:
            _check(self.lib.ya_sim_set_fixed(self._h, 0, 0), "set_fixed")
        else:
            _check(self.lib.ya_sim_set_fixed(self._h, 1, int(point_id)), "set_fixed")

    def set_fixed_xy(self, point_id):
        _check(self.lib.ya_sim_set_fixed(self._h, 2, int(point_id)), "set_fixed_xy")

    @property
    def cube_size(self):
        raise AttributeError("cube_size is write-only here")

    @cube_size.setter
    def cube_size(self, value):
        _check(self.lib.ya_sim_set_cube_size(self._h, float(value)), "set cube_size")

    # --- inits.cuh ---------------------------------------------------------
    def random_sphere(self, dist_to_nb, seed):
        _check(self.lib.ya_sim_random_sphere(self._h, float(dist_to_nb), int(seed)),
               "random_sphere")

    # --- state read-back for checks ---------------------------------------
    def positions(self):
        """copy_to_host() and a copy of h_X[:h_n]."""
        self.copy_to_host()
        return self.h_X[: self.h_n].copy()

    def old_v(self):
        out = np.empty((self.n_max, 3), dtype=np.float32)
        _check(self.lib.ya_sim_get_old_v(self._h, out.ctypes.data_as(C.POINTER(C.c_float))),
               "get_old_v")
        return out

    def set_old_v(self, v):
        v = np.ascontiguousarray(v, dtype=np.float32).reshape(self.n_max, 3)
        _check(self.lib.ya_sim_set_old_v(self._h, v.ctypes.data_as(C.POINTER(C.c_float))),
               "set_old_v")

    def _grid_buffers(self, gs):
        return (np.empty(self.n_max, np.int32), np.empty(self.n_max, np.int32),
                np.empty(gs ** 3, np.int32), np.empty(gs ** 3, np.int32))

    def grid(self):
        """cube_id, point_id, cube_start, cube_end of the solver's own Grid."""
        a, b, c, d = self._grid_buffers(self.grid_size)
        _check(self.lib.ya_sim_get_grid(self._h, _ip(a), _ip(b), _ip(c), _ip(d)), "get_grid")
        return a, b, c, d

    def build_grid(self, grid_size, cube_size=1.0):
        """Grid{n_max, grid_size}.build(points, cube_size) and its arrays."""
        a, b, c, d = self._grid_buffers(int(grid_size))
        _check(self.lib.ya_sim_build_grid(self._h, int(grid_size), float(cube_size),
                                          _ip(a), _ip(b), _ip(c), _ip(d)), "build_grid")
        return a, b, c, d

    # --- model extras --------------------------------------------------------
    def set_param(self, name, value):
        _check(self.lib.ya_sim_set_param(self._h, name.encode(), float(value)), "set_param")
        return 0

    def set_prop(self, name, values):
        v = np.ascontiguousarray(values, dtype=np.int32)
        _check(self.lib.ya_sim_set_prop(self._h, name.encode(), _ip(v), len(v)), "set_prop")

    def get_prop(self, name, n):
        v = np.empty(int(n), dtype=np.int32)
        _check(self.lib.ya_sim_get_prop(self._h, name.encode(), _ip(v), len(v)), "get_prop")
        return v

    def set_links(self, pairs, strength=0.2):
        ab = np.ascontiguousarray(pairs, dtype=np.int32).reshape(-1, 2)
        _check(self.lib.ya_sim_set_links(self._h, _ip(ab), len(ab), float(strength)), "set_links")

    def set_reduce_order(self, order):
        return self.lib.ya_sim_set_reduce_order(self._h, int(order))

    def profile(self, enable, every=1):
        """Time every `every`-th launch of the force kernel with HIP events."""
        return self.lib.ya_sim_profile(self._h, int(every) if enable else 0)

    def profile_read(self):
        ms = C.c_double()
        launches = C.c_int()
        _check(self.lib.ya_sim_profile_read(self._h, C.byref(ms), C.byref(launches)),
               "profile_read")
        return ms.value, launches.value


def models(lib=None):
    lib = lib if lib is not None else _ffi.device_lib()
    return [lib.ya_models_name(i).decode() for i in range(lib.ya_models_count())]
