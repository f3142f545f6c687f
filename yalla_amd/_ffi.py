"""ctypes binding of the model-harness C ABI (include/yalla_models.h).

`bind(path)` loads any shared library that implements that header and declares
the argument types of every entry point.  `device_lib()` loads the HIP build,
yalla_amd/libyalla_models.so (which pulls in libyalla_hip.so through its
$ORIGIN rpath) and raises if it is missing: the product has no CPU fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# YALLA_MODELS_LIB: another build of the same library (A/B of build flags, tools/gpu_*.sh)
DEVICE_LIB = os.environ.get("YALLA_MODELS_LIB") or os.path.join(_HERE, "libyalla_models.so")
DEVICE_LIB_FAST = os.path.join(_HERE, "libyalla_models_fast.so")  # the fast-arithmetic tier
CORE_LIB = os.path.join(_HERE, "libyalla_hip.so")

_pf = C.POINTER(C.c_float)
_pi = C.POINTER(C.c_int)
_sim = C.c_void_p

# name -> (restype, argtypes); mirrors include/yalla_models.h one to one.
MODELS_ABI = {
    "ya_models_is_device": (C.c_int, []),
    "ya_models_arith": (C.c_int, []),
    "ya_models_count": (C.c_int, []),
    "ya_models_name": (C.c_char_p, [C.c_int]),
    "ya_sim_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_float, C.POINTER(_sim)]),
    "ya_sim_destroy": (None, [_sim]),
    "ya_sim_n_floats": (C.c_int, [_sim]),
    "ya_sim_n_max": (C.c_int, [_sim]),
    "ya_sim_h_X": (_pf, [_sim]),
    "ya_sim_set_h_n": (C.c_int, [_sim, C.c_int]),
    "ya_sim_get_h_n": (C.c_int, [_sim]),
    "ya_sim_copy_to_device": (C.c_int, [_sim]),
    "ya_sim_copy_to_host": (C.c_int, [_sim]),
    "ya_sim_get_d_n": (C.c_int, [_sim]),
    "ya_sim_take_steps": (C.c_int, [_sim, C.c_float, C.c_int]),
    "ya_sim_synchronize": (C.c_int, [_sim]),
    "ya_sim_set_fixed": (C.c_int, [_sim, C.c_int, C.c_int]),
    "ya_sim_set_cube_size": (C.c_int, [_sim, C.c_float]),
    "ya_sim_random_sphere": (C.c_int, [_sim, C.c_float, C.c_uint]),
    "ya_sim_get_old_v": (C.c_int, [_sim, _pf]),
    "ya_sim_set_old_v": (C.c_int, [_sim, _pf]),
    "ya_sim_get_grid": (C.c_int, [_sim, _pi, _pi, _pi, _pi]),
    "ya_sim_build_grid": (C.c_int, [_sim, C.c_int, C.c_float, _pi, _pi, _pi, _pi]),
    "ya_sim_set_param": (C.c_int, [_sim, C.c_char_p, C.c_double]),
    "ya_sim_set_prop": (C.c_int, [_sim, C.c_char_p, _pi, C.c_int]),
    "ya_sim_get_prop": (C.c_int, [_sim, C.c_char_p, _pi, C.c_int]),
    "ya_sim_set_links": (C.c_int, [_sim, _pi, C.c_int, C.c_float]),
    "ya_sim_set_reduce_order": (C.c_int, [_sim, C.c_int]),
    "ya_slab_init": (C.c_int, [_sim, C.c_float, C.c_float, C.c_float, _pi]),
    "ya_slab_info": (C.c_long, [_sim, C.c_int]),
    "ya_slab_n_own": (C.c_int, [_sim]),
    "ya_slab_n_local": (C.c_int, [_sim]),
    "ya_slab_setup": (C.c_int, [_sim, C.c_int, C.c_int, C.c_int, C.c_int]),
    "ya_slab_set_transport": (C.c_int, [_sim, C.c_void_p, C.c_void_p, C.c_void_p]),
    "ya_slab_use_rccl": (C.c_int, [_sim, C.c_void_p]),
    "ya_slab_step": (C.c_int, [_sim, C.c_float, C.c_int]),
    "ya_slab_get_own": (C.c_int, [_sim, _pf, _pi]),
    "ya_slab_plan": (C.c_int, [_pf, C.c_int, C.c_int, C.c_int, C.c_float, _pf, _pi]),
    "ya_slab_decompose": (C.c_int, [_sim, _pf, C.c_int, C.c_int, C.c_int, C.c_float]),
    "ya_check_sqrt": (C.c_long, [C.c_uint, C.c_uint]),
    "ya_check_reciprocal": (C.c_long, [C.c_uint, C.c_uint]),
    "ya_sim_profile": (C.c_int, [_sim, C.c_int]),
    "ya_sim_profile_read": (C.c_int, [_sim, C.POINTER(C.c_double), _pi]),
}

# include/yalla_hip.h, for the export check (no compute calls without a GPU).
CORE_ABI = [
    "ya_abi_version", "ya_malloc", "ya_free", "ya_memset_async", "ya_memcpy_h2d",
    "ya_memcpy_d2h", "ya_host_alloc", "ya_host_free", "ya_memcpy_d2d_async", "ya_device_synchronize", "ya_get_n",
    "ya_grid_create", "ya_grid_destroy", "ya_grid_arrays", "ya_grid_offsets",
    "ya_grid_build", "ya_grid_build_sorted", "ya_grid_build_sorted_begin", "ya_grid_build_sorted_begin_publish",
    "ya_grid_build_sorted_finish", "ya_grid_rebuild_sorted", "ya_n_reader_create", "ya_n_reader_destroy",
    "ya_n_read_begin", "ya_n_read_end", "ya_grid_status", "ya_reduce_mean", "ya_reduce_sum_packed",
    "ya_reduce_workspace_bytes", "ya_select_z", "ya_select_workspace_bytes", "ya_gather_rows",
    "ya_gather_rows_pair",
    "ya_append_rows", "ya_comm_unique_id", "ya_comm_create", "ya_comm_create_loopback", "ya_comm_create_from_env",
    "ya_comm_destroy", "ya_comm_rank", "ya_comm_world", "ya_comm_exchange", "ya_comm_exchange_v",
    "ya_comm_allreduce_sum", "ya_comm_allreduce_host", "ya_comm_self_exchange", "ya_comm_info", "ya_reduce_partials",
    "ya_shader_clock_mhz", "ya_grid_set_cube_range", "ya_grid_forget_order", "ya_copy_component", "ya_pack_cells", "ya_append_cells", "ya_fill_holes", "ya_find_id", "ya_max_abs_diff", "ya_max_abs_diff_partials",
    "ya_slab_guard_update", "ya_slab_pack", "ya_async_read_create", "ya_async_read_destroy",
    "ya_async_read_begin", "ya_async_read_end", "ya_async_read_target", "ya_async_read_mark",
]


def bind(path):
    """Load `path` and type every include/yalla_models.h entry point."""
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()')")
    lib = C.CDLL(path, mode=C.RTLD_LOCAL)
    for name, (res, args) in MODELS_ABI.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    return lib


_device = {}


def device_lib(arith="exact"):
    """The HIP engine.  Raises if the extension has not been built.  arith "exact" (default:
    bit-comparable with the oracle) or "fast" (libyalla_models_fast.so: contracted multiply-adds,
    bare v_sqrt_f32 / v_rcp_f32; within 1e-5 relative of the exact tier)."""
    if arith not in ("exact", "fast"):
        raise ValueError("arith must be 'exact' or 'fast'")
    if arith not in _device:
        path = DEVICE_LIB if arith == "exact" else DEVICE_LIB_FAST
        lib = bind(path)
        if lib.ya_models_is_device() != 1 or lib.ya_models_arith() != (arith == "fast"):
            raise RuntimeError(f"{path} is not the {arith}-arithmetic HIP build")
        _device[arith] = lib
    return _device[arith]
