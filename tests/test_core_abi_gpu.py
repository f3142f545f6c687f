"""Direct calls into libyalla_hip.so (include/yalla_hip.h) on the GPU: grid build
against a numpy restatement (bit-exact), ordered selection, row gather, the
deterministic reduction, status flag."""
import ctypes as C

import numpy as np
import pytest

from yalla_amd import _ffi

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip():
    lib = C.CDLL(_ffi.CORE_LIB, mode=C.RTLD_LOCAL)
    vp, sz, i32, f32 = C.c_void_p, C.c_size_t, C.c_int, C.c_float
    lib.ya_malloc.argtypes = [C.POINTER(vp), sz]
    lib.ya_free.argtypes = [vp]
    lib.ya_memcpy_h2d.argtypes = [vp, vp, sz]
    lib.ya_memcpy_d2h.argtypes = [vp, vp, sz]
    lib.ya_grid_create.argtypes = [i32, i32, C.POINTER(vp)]
    lib.ya_grid_destroy.argtypes = [vp]
    lib.ya_grid_arrays.argtypes = [vp] + [C.POINTER(vp)] * 4
    lib.ya_grid_offsets.argtypes = [vp, C.POINTER(vp)]
    lib.ya_grid_build.argtypes = [vp, vp, sz, i32, f32, vp]
    lib.ya_grid_status.argtypes = [vp, C.POINTER(i32), i32]
    lib.ya_grid_build_sorted.argtypes = [vp, vp, sz, vp, i32, f32, vp, sz, vp, vp]
    lib.ya_grid_rebuild_sorted.argtypes = [vp, vp, sz, sz, vp, i32, f32, vp, vp, vp]
    lib.ya_select_z.argtypes = [vp, sz, i32, f32, f32, vp, vp, vp, vp]
    lib.ya_select_workspace_bytes.restype = sz
    lib.ya_select_workspace_bytes.argtypes = [i32]
    lib.ya_gather_rows.argtypes = [vp, sz, vp, vp, i32, vp, vp]
    lib.ya_gather_rows_pair.argtypes = [vp, sz, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.ya_reduce_sum_packed.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.ya_reduce_mean.argtypes = [vp, i32, i32, vp, vp, vp]
    lib.ya_reduce_partials.argtypes = [vp, i32, i32, vp, C.POINTER(C.c_int), vp]
    lib.ya_reduce_workspace_bytes.restype = sz
    lib.ya_reduce_workspace_bytes.argtypes = [i32]
    lib.ya_device_synchronize.argtypes = []
    return lib


class Dev:
    def __init__(self, hip, array=None, nbytes=None):
        self.hip = hip
        self.p = C.c_void_p()
        nbytes = array.nbytes if array is not None else nbytes
        assert hip.ya_malloc(C.byref(self.p), max(nbytes, 4)) == 0
        if array is not None:
            a = np.ascontiguousarray(array)
            assert hip.ya_memcpy_h2d(self.p, a.ctypes.data, a.nbytes) == 0

    def get(self, dtype, count):
        out = np.empty(count, dtype)
        assert self.hip.ya_memcpy_d2h(out.ctypes.data, self.p, out.nbytes) == 0
        return out

    def __del__(self):
        self.hip.ya_free(self.p)


def numpy_grid(X, cs, gs):
    """solvers.cuh:349-378 in numpy, float32 arithmetic in the reference's order."""
    f = np.float32
    half = f(gs // 2)
    fx = np.floor(X[:, 0] / f(cs)) + half
    fy = (np.floor(X[:, 1] / f(cs)) + half) * f(gs)
    fz = ((np.floor(X[:, 2] / f(cs)) + half) * f(gs)) * f(gs)
    ids = ((fx + fy) + fz).astype(np.int32)
    order = np.argsort(ids, kind="stable").astype(np.int32)
    sorted_ids = ids[order]
    start = np.full(gs ** 3, -1, np.int32)
    end = np.full(gs ** 3, -2, np.int32)
    if len(ids) == 0:
        return sorted_ids, order, start, end
    first = np.r_[True, sorted_ids[1:] != sorted_ids[:-1]]
    last = np.r_[sorted_ids[1:] != sorted_ids[:-1], True]
    start[sorted_ids[first]] = np.nonzero(first)[0]
    end[sorted_ids[last]] = np.nonzero(last)[0]
    return sorted_ids, order, start, end


@pytest.mark.parametrize("n,stride_f,gs,cs", [(1, 3, 8, 1.0), (1000, 3, 20, 1.0), (50000, 5, 40, 0.7),
                                             (200000, 7, 64, 1.0)])
def test_grid_build_is_bit_exact(hip, n, stride_f, gs, cs):
    rng = np.random.default_rng(n)
    X = np.zeros((n, stride_f), np.float32)
    X[:, :3] = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32((gs - 3) * cs)
    X[:, 3:] = 7.0  # extra fields must be ignored
    dX = Dev(hip, X)
    g = C.c_void_p()
    assert hip.ya_grid_create(n, gs, C.byref(g)) == 0
    ptrs = [C.c_void_p() for _ in range(4)]
    hip.ya_grid_arrays(g, *[C.byref(p) for p in ptrs])
    offs = C.c_void_p()
    hip.ya_grid_offsets(g, C.byref(offs))
    for rebuild in range(3):  # later builds visit cells in the previous sorted order
        assert hip.ya_grid_build(g, dX.p, stride_f * 4, n, cs, None) == 0
        hip.ya_device_synchronize()
        got = []
        for p, count in zip(ptrs, (n, n, gs ** 3, gs ** 3)):
            out = np.empty(count, np.int32)
            hip.ya_memcpy_d2h(out.ctypes.data, p, out.nbytes)
            got.append(out)
        ref = numpy_grid(X, cs, gs)
        for name, a, b in zip(("cube_id", "point_id", "cube_start", "cube_end"), ref, got):
            assert np.array_equal(a, b), (name, rebuild)
        o = np.empty(gs ** 3 + 1, np.int32)
        hip.ya_memcpy_d2h(o.ctypes.data, offs, o.nbytes)
        counts = np.bincount(ref[0], minlength=gs ** 3)
        assert np.array_equal(o, np.r_[0, np.cumsum(counts)].astype(np.int32))
    bits = C.c_int(-1)
    assert hip.ya_grid_status(g, C.byref(bits), 0) == 0 and bits.value == 0
    hip.ya_grid_destroy(g)


def test_grid_build_with_changing_population(hip):
    """The visit order of a build comes from the previous build; the population may shrink
    (a slab's ghost layer), grow (proliferation) or collapse in between."""
    n_max, gs, cs, stride_f = 30000, 32, 1.0, 3
    rng = np.random.default_rng(5)
    X = ((rng.random((n_max, 3), dtype=np.float32) - 0.5) * np.float32(gs - 3)).astype(np.float32)
    dX = Dev(hip, X)
    g = C.c_void_p()
    assert hip.ya_grid_create(n_max, gs, C.byref(g)) == 0
    ptrs = [C.c_void_p() for _ in range(4)]
    hip.ya_grid_arrays(g, *[C.byref(p) for p in ptrs])
    for n in (20000, 19990, 19000, 30000, 29999, 12000, 500, 777, 0, 30000):
        assert hip.ya_grid_build(g, dX.p, stride_f * 4, n, cs, None) == 0
        hip.ya_device_synchronize()
        ref = numpy_grid(X[:n], cs, gs)
        for name, a, p, count in zip(("cube_id", "point_id", "cube_start", "cube_end"), ref, ptrs,
                                     (n, n, gs ** 3, gs ** 3)):
            out = np.empty(count, np.int32)
            hip.ya_memcpy_d2h(out.ctypes.data, p, out.nbytes)
            assert np.array_equal(a, out), (name, n)
    hip.ya_grid_destroy(g)


def test_count_published_by_the_binning_kernel(hip):
    """ya_grid_build_sorted_begin_publish (round 6): the first build of a step bins with the count read on the
    device AND hands that count to the host itself -- ya_n_read_end returns it without a copy in the stream -- for a
    population that changes between builds (proliferation), n = 0 included; _finish then gives numpy's arrays."""
    n_max, gs, cs = 20000, 32, 1.0
    vp, i32, f32c, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    hip.ya_n_reader_create.argtypes = [C.POINTER(vp)]
    hip.ya_n_reader_destroy.argtypes = [vp]
    hip.ya_n_read_end.argtypes = [vp, C.POINTER(i32)]
    hip.ya_grid_build_sorted_begin_publish.argtypes = [vp, vp, sz, vp, i32, f32c, vp, vp]
    hip.ya_grid_build_sorted_finish.argtypes = [vp, vp, sz, vp, i32, vp, sz, vp, vp]
    rng = np.random.default_rng(11)
    X = ((rng.random((n_max, 3), dtype=np.float32) - 0.5) * np.float32(gs - 3)).astype(np.float32)
    dX, dv = Dev(hip, X), Dev(hip, np.zeros((n_max, 3), np.float32))
    sorted_X, sorted_v = Dev(hip, nbytes=n_max * 16), Dev(hip, nbytes=n_max * 16)
    g, reader = vp(), vp()
    assert hip.ya_grid_create(n_max, gs, C.byref(g)) == 0 and hip.ya_n_reader_create(C.byref(reader)) == 0
    ptrs = [vp() for _ in range(4)]
    hip.ya_grid_arrays(g, *[C.byref(q) for q in ptrs])
    for n in (12000, 12001, 15000, 20000, 300, 0, 7777):
        d_n = Dev(hip, np.array([n], np.int32))
        assert hip.ya_grid_build_sorted_begin_publish(g, dX.p, 12, d_n.p, n_max, cs, reader, None) == 0
        got = i32(-1)
        assert hip.ya_n_read_end(reader, C.byref(got)) == 0 and got.value == n
        if n == 0:
            continue
        assert hip.ya_grid_build_sorted_finish(g, dX.p, 12, dv.p, n, sorted_X.p, 16, sorted_v.p, None) == 0
        hip.ya_device_synchronize()
        ref = numpy_grid(X[:n], cs, gs)
        for name, a, q, count in zip(("cube_id", "point_id", "cube_start", "cube_end"), ref, ptrs, (n, n, gs ** 3, gs ** 3)):
            out = np.empty(count, np.int32)
            hip.ya_memcpy_d2h(out.ctypes.data, q, out.nbytes)
            assert np.array_equal(a, out), (name, n)
    hip.ya_n_reader_destroy(reader)
    hip.ya_grid_destroy(g)


@pytest.mark.parametrize("point_f,entry_f", [(3, 4), (4, 8), (5, 6)])
def test_rebuild_from_sorted_cells_equals_build_from_original_order(hip, point_f, entry_f):
    """ya_grid_rebuild_sorted (second Heun stage): cells that sit in an earlier build's
    sorted arrays and have moved since give exactly the arrays a build from the
    original-order arrays gives -- public arrays, sorted entries and sorted old_v."""
    n, gs, cs = 40000, 32, 1.0
    rng = np.random.default_rng(point_f)
    X = np.zeros((n, point_f), np.float32)
    X[:, :3] = (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(gs - 4)
    X[:, 3:] = rng.random((n, point_f - 3), dtype=np.float32)
    V = rng.random((n, 3), dtype=np.float32)
    g = C.c_void_p()
    assert hip.ya_grid_create(n, gs, C.byref(g)) == 0
    ptrs = [C.c_void_p() for _ in range(4)]
    hip.ya_grid_arrays(g, *[C.byref(p) for p in ptrs])
    dV = Dev(hip, V)
    first, first_v = Dev(hip, nbytes=n * entry_f * 4), Dev(hip, nbytes=n * 16)
    assert hip.ya_grid_build_sorted(g, Dev(hip, X).p, point_f * 4, dV.p, n, cs, first.p, entry_f * 4,
                                    first_v.p, None) == 0
    entries = first.get(np.float32, n * entry_f).reshape(n, entry_f)
    ids = entries[:, point_f].view(np.int32)
    assert np.array_equal(np.sort(ids), np.arange(n))
    # the cells move (a predictor step); the sorted copy is updated in place
    X2 = X.copy()
    X2[:, :3] += (rng.random((n, 3), dtype=np.float32) - 0.5) * np.float32(0.8)
    entries[:, :point_f] = X2[ids]
    moved = Dev(hip, entries)
    out, out_v = Dev(hip, nbytes=n * entry_f * 4), Dev(hip, nbytes=n * 16)
    assert hip.ya_grid_rebuild_sorted(g, moved.p, entry_f * 4, point_f * 4, first_v.p, n, cs, out.p,
                                      out_v.p, None) == 0
    hip.ya_device_synchronize()
    got = []
    for p, count in zip(ptrs, (n, n, gs ** 3, gs ** 3)):
        a = np.empty(count, np.int32)
        hip.ya_memcpy_d2h(a.ctypes.data, p, a.nbytes)
        got.append(a)
    got_entries = out.get(np.float32, n * entry_f).reshape(n, entry_f)
    got_v = out_v.get(np.float32, n * 4).reshape(n, 4)
    # reference: a build from the original-order arrays
    want, want_v = Dev(hip, nbytes=n * entry_f * 4), Dev(hip, nbytes=n * 16)
    assert hip.ya_grid_build_sorted(g, Dev(hip, X2).p, point_f * 4, dV.p, n, cs, want.p, entry_f * 4,
                                    want_v.p, None) == 0
    hip.ya_device_synchronize()
    ref = numpy_grid(X2, cs, gs)
    for name, a, b in zip(("cube_id", "point_id", "cube_start", "cube_end"), ref, got):
        assert np.array_equal(a, b), name
    want_entries = want.get(np.float32, n * entry_f).reshape(n, entry_f)
    assert np.array_equal(got_entries[:, :point_f + 1].view(np.uint32),
                          want_entries[:, :point_f + 1].view(np.uint32))
    assert np.array_equal(got_v[:, :3], want_v.get(np.float32, n * 4).reshape(n, 4)[:, :3])
    assert np.array_equal(got_v[:, :3], V[ref[1]])
    hip.ya_grid_destroy(g)


def test_out_of_grid_is_flagged(hip):
    X = np.array([[0, 0, 0], [0, 0, 100.0]], np.float32)  # linear id beyond gs^3
    dX = Dev(hip, X)
    g = C.c_void_p()
    assert hip.ya_grid_create(2, 10, C.byref(g)) == 0
    assert hip.ya_grid_build(g, dX.p, 12, 2, 1.0, None) == 0
    bits = C.c_int(0)
    assert hip.ya_grid_status(g, C.byref(bits), 1) == 0 and bits.value == 1
    assert hip.ya_grid_status(g, C.byref(bits), 0) == 0 and bits.value == 0  # cleared
    hip.ya_grid_destroy(g)


@pytest.mark.parametrize("n", [0, 1, 2047, 2048, 2049, 100000])
def test_select_and_gather_keep_order(hip, n):
    rng = np.random.default_rng(5)
    X = rng.random((max(n, 1), 4), dtype=np.float32)
    dX = Dev(hip, X)
    d_idx = Dev(hip, nbytes=4 * max(n, 1))
    d_count = Dev(hip, nbytes=16)
    d_ws = Dev(hip, nbytes=hip.ya_select_workspace_bytes(max(n, 1)))
    assert hip.ya_select_z(dX.p, 16, n, 0.25, 0.6, d_idx.p, d_count.p, d_ws.p, None) == 0
    ref = np.nonzero((X[:n, 2] >= np.float32(0.25)) & (X[:n, 2] < np.float32(0.6)))[0].astype(np.int32)
    count = int(d_count.get(np.int32, 1)[0])
    assert count == len(ref)
    assert np.array_equal(d_idx.get(np.int32, max(n, 1))[:count], ref)
    cap = max(count, 1)
    d_out = Dev(hip, nbytes=16 * cap)
    assert hip.ya_gather_rows(dX.p, 16, d_idx.p, d_count.p, cap, d_out.p, None) == 0
    assert np.array_equal(d_out.get(np.float32, 4 * cap).reshape(cap, 4)[:count], X[ref])


def test_reduce_sum_packed_is_the_sum_of_reduce_mean(hip):
    """ya_reduce_sum_packed: bit for bit the sum ya_reduce_mean leaves beside its mean, followed by
    the count in two pieces that survive a float all-reduce exactly."""
    rng = np.random.default_rng(2)
    for n, nf in ((1, 3), (70001, 3), (1234567, 3), (300000, 7)):
        v = (rng.random((n, nf), dtype=np.float32) - 0.5).astype(np.float32)
        dv = Dev(hip, v)
        d_ws = Dev(hip, nbytes=hip.ya_reduce_workspace_bytes(nf))
        d_a, d_b = Dev(hip, nbytes=8 * nf), Dev(hip, nbytes=4 * (nf + 2))
        assert hip.ya_reduce_mean(dv.p, nf, n, d_a.p, d_ws.p, None) == 0
        assert hip.ya_reduce_sum_packed(dv.p, nf, n, d_b.p, d_ws.p, None) == 0
        a, b = d_a.get(np.float32, 2 * nf), d_b.get(np.float32, nf + 2)
        assert np.array_equal(a[nf:].view(np.uint32), b[:nf].view(np.uint32))
        assert b[nf] == n % 4096 and b[nf + 1] == n // 4096


def test_gather_rows_pair_is_two_gathers(hip):
    """ya_gather_rows_pair against numpy: two index lists into one array in one launch, counts
    read on the device and clamped to the capacity, a NULL list skipped."""
    rng = np.random.default_rng(11)
    n, cap = 50000, 7000
    src = rng.random((n, 3), dtype=np.float32)
    idx0 = rng.permutation(n)[:6000].astype(np.int32)
    idx1 = rng.permutation(n)[:9000].astype(np.int32)   # longer than the capacity: clamped
    d_src, d_i0, d_i1 = Dev(hip, src), Dev(hip, idx0), Dev(hip, idx1)
    d_counts = Dev(hip, np.array([len(idx0), len(idx1), 0, 0], np.int32))
    d_o0, d_o1 = Dev(hip, np.zeros((cap, 3), np.float32)), Dev(hip, np.zeros((cap, 3), np.float32))
    assert hip.ya_gather_rows_pair(d_src.p, 12, d_i0.p, d_counts.p, d_o0.p, d_i1.p, C.c_void_p(d_counts.p.value + 4), d_o1.p,
                                   cap, None) == 0
    o0 = d_o0.get(np.float32, 3 * cap).reshape(cap, 3)
    o1 = d_o1.get(np.float32, 3 * cap).reshape(cap, 3)
    assert np.array_equal(o0[:6000], src[idx0]) and not o0[6000:].any()
    assert np.array_equal(o1, src[idx1[:cap]])
    d_o1b = Dev(hip, np.zeros((cap, 3), np.float32))
    assert hip.ya_gather_rows_pair(d_src.p, 12, None, None, None, d_i1.p, C.c_void_p(d_counts.p.value + 4), d_o1b.p, cap, None) == 0
    assert np.array_equal(d_o1b.get(np.float32, 3 * cap).reshape(cap, 3), src[idx1[:cap]])


@pytest.mark.parametrize("n,nf", [(1, 3), (255, 3), (70000, 5), (300000, 7)])
def test_reduce_mean_matches_its_documented_order(hip, n, nf):
    rng = np.random.default_rng(1)
    v = (rng.random((n, nf), dtype=np.float32) - 0.5).astype(np.float32)
    dv = Dev(hip, v)
    d_out = Dev(hip, nbytes=8 * nf)
    d_ws = Dev(hip, nbytes=hip.ya_reduce_workspace_bytes(nf))
    assert hip.ya_reduce_mean(dv.p, nf, n, d_out.p, d_ws.p, None) == 0
    out = d_out.get(np.float32, 2 * nf)
    # DESIGN.md "deterministic COM reduction", in numpy float32
    B = min(max((n + 255) // 256, 1), 1024)
    pad = np.zeros((B * 256 * ((n + B * 256 - 1) // (B * 256)), nf), np.float32)
    pad[:n] = v
    lanes = pad.reshape(-1, B, 256, nf)
    acc = np.zeros((B, 256, nf), np.float32)
    for chunk in lanes:
        acc = acc + chunk
    def fold(a):  # a: (..., 256, nf)
        s = 128
        while s >= 1:
            a = a[..., :s, :] + a[..., s:2 * s, :]
            s //= 2
        return a[..., 0, :]
    part = fold(acc)                       # (B, nf)
    lanes2 = np.zeros((256 * ((B + 255) // 256), nf), np.float32)
    lanes2[:B] = part
    acc2 = np.zeros((256, nf), np.float32)
    for chunk in lanes2.reshape(-1, 256, nf):
        acc2 = acc2 + chunk
    total = fold(acc2)
    assert np.array_equal(out[nf:].view(np.uint32), total.view(np.uint32))
    inv = np.float32(1.0 / np.float64(np.float32(n)))
    assert np.array_equal(out[:nf].view(np.uint32), (total * inv).view(np.uint32))


@pytest.mark.parametrize("n,nf", [(1, 3), (70000, 5), (1234567, 3)])
def test_reduce_partials_are_the_first_half_of_reduce_mean(hip, n, nf):
    """ya_reduce_partials (round 5: the update kernels fold these themselves): B = clamp(ceil(n / 256), 1, 1024)
    per-block sums in the documented order; folded as ya::fixed_velocity_from_partials folds them they give the
    bits ya_reduce_mean leaves."""
    rng = np.random.default_rng(5)
    v = (rng.random((n, nf), dtype=np.float32) - 0.5).astype(np.float32)
    dv = Dev(hip, v)
    d_ws = Dev(hip, nbytes=hip.ya_reduce_workspace_bytes(nf))
    count = C.c_int(0)
    assert hip.ya_reduce_partials(dv.p, nf, n, d_ws.p, C.byref(count), None) == 0
    B = min(max((n + 255) // 256, 1), 1024)
    assert count.value == B
    assert hip.ya_device_synchronize() == 0
    part = d_ws.get(np.float32, B * nf).reshape(B, nf)

    def fold(a):  # a: (..., 256, nf)
        s = 128
        while s >= 1:
            a = a[..., :s, :] + a[..., s:2 * s, :]
            s //= 2
        return a[..., 0, :]
    pad = np.zeros((B * 256 * ((n + B * 256 - 1) // (B * 256)), nf), np.float32)
    pad[:n] = v
    acc = np.zeros((B, 256, nf), np.float32)
    for chunk in pad.reshape(-1, B, 256, nf):
        acc = acc + chunk
    assert np.array_equal(part.view(np.uint32), fold(acc).view(np.uint32))
    # ... and the second half, as the update kernels do it
    lanes = np.zeros((256 * ((B + 255) // 256), nf), np.float32)
    lanes[:B] = part
    acc2 = np.zeros((256, nf), np.float32)
    for chunk in lanes.reshape(-1, 256, nf):
        acc2 = acc2 + chunk
    d_out = Dev(hip, nbytes=8 * nf)
    assert hip.ya_reduce_mean(dv.p, nf, n, d_out.p, d_ws.p, None) == 0
    out = d_out.get(np.float32, 2 * nf)
    assert np.array_equal(out[nf:].view(np.uint32), fold(acc2).view(np.uint32))


# ---- round 4 entry points ---------------------------------------------------------------------------
@pytest.fixture(scope="module")
def hip4(hip):
    vp, sz, i32, f32, f64 = C.c_void_p, C.c_size_t, C.c_int, C.c_float, C.c_double
    three_p, three_s = vp * 3, sz * 3
    hip.ya_grid_set_cube_range.argtypes = [vp, i32, i32]
    hip.ya_grid_forget_order.argtypes = [vp]
    hip.ya_pack_cells.argtypes = [three_p, three_s, vp, vp, i32, vp, sz, vp]
    hip.ya_append_cells.argtypes = [three_p, three_s, i32, vp, vp, i32, sz, vp, vp, vp]
    hip.ya_fill_holes.argtypes = [three_p, three_s, vp, vp, vp, vp, vp, vp, i32, i32, vp]
    hip.ya_copy_component.argtypes = [vp, sz, i32, i32, vp, vp]
    hip.ya_find_id.argtypes = [vp, i32, i32, vp, vp]
    hip.ya_max_abs_diff.argtypes = [vp, sz, vp, sz, i32, f32, f32, f32, vp, vp]
    hip.ya_max_abs_diff_partials.argtypes = [i32]
    hip.ya_slab_guard_update.argtypes = [vp, i32, vp, i32, f32, f32, vp, vp]
    hip.ya_slab_pack.argtypes = [vp, i32, i32, vp, vp, vp, i32, vp, i32, f32, f32, vp, i32, i32, i32, vp, vp]
    hip.ya_async_read_create.argtypes = [sz, C.POINTER(vp)]
    hip.ya_async_read_destroy.argtypes = [vp]
    hip.ya_async_read_begin.argtypes = [vp, vp, vp]
    hip.ya_async_read_end.argtypes = [vp, vp]
    hip.ya_shader_clock_mhz.argtypes = [f64, C.POINTER(f64)]
    hip.ya_comm_create_loopback.argtypes = [i32, C.POINTER(vp)]
    hip.ya_comm_destroy.argtypes = [vp]
    hip.ya_comm_allreduce_sum.argtypes = [vp, vp, i32, vp]
    hip.ya_comm_exchange_v.argtypes = [vp, vp, sz, vp, sz, vp, sz, vp, sz, vp]
    return hip


def _grid_arrays(hip, g, n, gs):
    ptr = [C.c_void_p() for _ in range(4)]
    assert hip.ya_grid_arrays(g, *[C.byref(p) for p in ptr]) == 0
    offs = C.c_void_p()
    assert hip.ya_grid_offsets(g, C.byref(offs)) == 0
    sizes = [n, n, gs ** 3, gs ** 3]
    out = []
    for p, m in zip(ptr, sizes):
        a = np.empty(m, np.int32)
        assert hip.ya_memcpy_d2h(a.ctypes.data, p, a.nbytes) == 0
        out.append(a)
    o = np.empty(gs ** 3 + 1, np.int32)
    assert hip.ya_memcpy_d2h(o.ctypes.data, offs, o.nbytes) == 0
    return out + [o]


def test_cube_range_builds_equal_full_builds_and_strays_are_flagged(hip4):
    """ya_grid_set_cube_range: cells in a few z-planes of a 40^3 grid; after one build that scans every tile the
    prefix sum runs over the promised planes only -- same five arrays as a grid without the promise, build after
    build with moving cells; a cell outside the promise raises YA_STATUS_OUT_OF_RANGE and is kept inside."""
    hip = hip4
    gs, n = 40, 30000
    rng = np.random.default_rng(11)
    X = np.empty((n, 3), np.float32)
    X[:, :2] = rng.uniform(-15, 15, (n, 2))
    X[:, 2] = rng.uniform(2.0, 7.9, n)                      # planes 22 .. 27 (gs / 2 = 20)
    plain, ranged = C.c_void_p(), C.c_void_p()
    assert hip.ya_grid_create(n, gs, C.byref(plain)) == 0 and hip.ya_grid_create(n, gs, C.byref(ranged)) == 0
    assert hip.ya_grid_set_cube_range(ranged, 21 * gs * gs, 29 * gs * gs) == 0
    d = Dev(hip, X)
    for step in range(4):
        if step:
            X[:, :2] += rng.normal(0, 0.3, (n, 2)).astype(np.float32)
            X[:, 2] = np.clip(X[:, 2] + rng.normal(0, 0.1, n).astype(np.float32), 1.1, 8.9)
            assert hip.ya_memcpy_h2d(d.p, X.ctypes.data, X.nbytes) == 0
        for g in (plain, ranged):
            assert hip.ya_grid_build(g, d.p, 12, n, 1.0, None) == 0
        for a, b in zip(_grid_arrays(hip, plain, n, gs), _grid_arrays(hip, ranged, n, gs)):
            assert np.array_equal(a, b), step
        bits = C.c_int(-1)
        assert hip.ya_grid_status(ranged, C.byref(bits), 1) == 0 and bits.value == 0
    X[7, 2] = 12.5                                           # plane 32: outside the promise
    assert hip.ya_memcpy_h2d(d.p, X.ctypes.data, X.nbytes) == 0
    assert hip.ya_grid_build(ranged, d.p, 12, n, 1.0, None) == 0
    bits = C.c_int(0)
    assert hip.ya_grid_status(ranged, C.byref(bits), 1) == 0 and bits.value == 2   # YA_STATUS_OUT_OF_RANGE
    ids = _grid_arrays(hip, ranged, n, gs)[0]
    assert ids.min() >= 21 * gs * gs - 2048 and ids.max() < 29 * gs * gs + 2048     # kept inside (whole scan tiles)
    hip.ya_grid_destroy(plain)
    hip.ya_grid_destroy(ranged)


def test_cell_records_packed_appended_and_holes_filled(hip4):
    """ya_pack_cells / ya_append_cells / ya_fill_holes against numpy: three arrays with their own row widths
    moved by one launch each (a slab's {point, old_v, global id})."""
    hip = hip4
    rng = np.random.default_rng(5)
    n, cap, header = 5000, 900, 16
    rows = [5, 3, 1]                                          # floats per row: Po_cell, old_v, id
    fields = [rng.random((n + 2 * cap, r)).astype(np.float32) for r in rows]
    dev = [Dev(hip, f) for f in fields]
    arrays = (C.c_void_p * 3)(*[d.p for d in dev])
    sizes = (C.c_size_t * 3)(*[4 * r for r in rows])
    idx = np.sort(rng.choice(n, 700, replace=False)).astype(np.int32)
    d_idx, d_count = Dev(hip, idx), Dev(hip, np.array([len(idx)], np.int32))
    msg_bytes = header + cap * 4 * sum(rows)
    message = Dev(hip, nbytes=msg_bytes)
    assert hip.ya_pack_cells(arrays, sizes, d_idx.p, d_count.p, cap, message.p, header, None) == 0
    raw = message.get(np.uint8, msg_bytes)
    assert raw[:4].view(np.int32)[0] == len(idx)
    at = header
    for f, r in zip(fields, rows):
        got = raw[at:at + 4 * r * len(idx)].view(np.float32).reshape(-1, r)
        assert np.array_equal(got, f[idx])
        at += cap * 4 * r
    # the same message appended twice (as "lower" and "upper") behind n rows
    d_n, d_counts = Dev(hip, nbytes=4), Dev(hip, nbytes=8)
    assert hip.ya_append_cells(arrays, sizes, n, message.p, message.p, cap, header, d_n.p, d_counts.p, None) == 0
    assert d_n.get(np.int32, 1)[0] == n + 2 * len(idx) and list(d_counts.get(np.int32, 2)) == [len(idx)] * 2
    for d, f, r in zip(dev, fields, rows):
        got = d.get(np.float32, (n + 2 * cap) * r).reshape(-1, r)
        assert np.array_equal(got[n:n + len(idx)], f[idx]) and np.array_equal(got[n + len(idx):n + 2 * len(idx)], f[idx])
        assert np.array_equal(got[:n], f[:n])
    # holes: cells idx_lo and idx_hi leave; the staying cells of the tail move into the holes below n_new
    leave = np.sort(rng.choice(n, 600, replace=False)).astype(np.int32)
    lo, hi = leave[::2].copy(), leave[1::2].copy()
    n_new = n - len(leave)
    stay_tail = np.setdiff1d(np.arange(n_new, n), leave).astype(np.int32)
    movers = (stay_tail - n_new).astype(np.int32)
    holes = np.concatenate([lo[lo < n_new], hi[hi < n_new]])
    assert len(holes) == len(movers)
    d_lo, d_hi, d_mv = Dev(hip, lo), Dev(hip, hi), Dev(hip, movers if len(movers) else np.zeros(1, np.int32))
    c_lo, c_hi, c_mv = (Dev(hip, np.array([len(a)], np.int32)) for a in (lo, hi, movers))
    before = [d.get(np.float32, n * r).reshape(-1, r) for d, r in zip(dev, rows)]
    assert hip.ya_fill_holes(arrays, sizes, d_lo.p, c_lo.p, d_hi.p, c_hi.p, d_mv.p, c_mv.p, n_new, n - n_new, None) == 0
    for d, b, r in zip(dev, before, rows):
        want = b.copy()
        want[holes] = b[stay_tail]
        got = d.get(np.float32, n * r).reshape(-1, r)
        assert np.array_equal(got[:n_new], want[:n_new])
    # what is left in the first n_new rows is exactly the staying cells
    stay = np.setdiff1d(np.arange(n), leave)
    got0 = dev[0].get(np.float32, n * rows[0]).reshape(-1, rows[0])[:n_new]
    assert sorted(map(tuple, got0.tolist())) == sorted(map(tuple, before[0][stay].tolist()))


def test_drift_guard_and_payload(hip4):
    """ya_copy_component, ya_find_id, ya_max_abs_diff (weights by band), ya_slab_guard_update and ya_slab_pack
    (the packed sum of ya_reduce_sum_packed, the guard folded in the same kernel, votes, the fixed point)."""
    hip = hip4
    rng = np.random.default_rng(9)
    n, nw = 40000, 5
    X = rng.normal(0, 3, (n, nw)).astype(np.float32)
    moved = X.copy()
    moved[:, 2] += rng.normal(0, 0.01, n).astype(np.float32)
    dX, dM = Dev(hip, X), Dev(hip, moved)
    z = Dev(hip, nbytes=4 * n)
    assert hip.ya_copy_component(dX.p, 4 * nw, 2, n, z.p, None) == 0
    assert np.array_equal(z.get(np.float32, n), X[:, 2])
    ids = rng.permutation(n).astype(np.int32)
    d_ids, d_at = Dev(hip, ids), Dev(hip, nbytes=4)
    assert hip.ya_find_id(d_ids.p, n, int(ids[1234]), d_at.p, None) == 0 and d_at.get(np.int32, 1)[0] == 1234
    assert hip.ya_find_id(d_ids.p, n, n + 5, d_at.p, None) == 0 and d_at.get(np.int32, 1)[0] == -1
    parts = hip.ya_max_abs_diff_partials(n)
    partial = Dev(hip, np.zeros(1024, np.float32))
    lo_face, hi_face, width = -1.0, 2.0, 1.375
    z_of_moved = C.c_void_p(dM.p.value + 8)
    assert hip.ya_max_abs_diff(z_of_moved, 4 * nw, z.p, 4, n, lo_face, hi_face, width, partial.p, None) == 0
    w = np.where((np.abs(X[:, 2] - lo_face) <= width) | (np.abs(X[:, 2] - hi_face) <= width), 1.0, 0.5).astype(np.float32)
    want = (np.abs(moved[:, 2] - X[:, 2]) * w).max()
    assert partial.get(np.float32, parts).max() == want
    pred = Dev(hip, np.full(256, 0.01, np.float32))
    state = Dev(hip, np.zeros(4, np.float32))
    limit, lag = 0.125, 2.5
    assert hip.ya_slab_guard_update(partial.p, parts, pred.p, 256, limit, lag, state.p, None) == 0
    st = state.get(np.float32, 4)
    assert st[0] == want and st[1] == np.float32(0.01)
    assert st[2] == float(want + np.float32(lag) * np.float32(0.01) > limit) and st[3] == float(want + np.float32(0.01) > limit)
    assert (partial.get(np.float32, parts) == 0).all(), "the partials are left zeroed"
    # the payload: sum + count pieces as ya_reduce_sum_packed, then votes and the fixed point's row
    ws = Dev(hip, nbytes=hip.ya_reduce_workspace_bytes(nw))
    packed, payload = Dev(hip, nbytes=4 * (nw + 2)), Dev(hip, np.full(nw + 8, -7, np.float32))
    assert hip.ya_reduce_sum_packed(dX.p, nw, n, packed.p, ws.p, None) == 0
    fix = Dev(hip, np.array([321], np.int32))
    big = Dev(hip, np.full(256, 0.2, np.float32))             # a predictor step beyond the limit: error
    assert hip.ya_slab_pack(dX.p, nw, n, payload.p, ws.p, partial.p, parts, big.p, 256, limit, lag, state.p, 1, 1, 0,
                            fix.p, None) == 0
    out = payload.get(np.float32, nw + 8)
    assert np.array_equal(out[:nw + 2].view(np.uint32), packed.get(np.float32, nw + 2).view(np.uint32))
    assert out[nw + 2] == 1 and out[nw + 3] == 1 and np.array_equal(out[nw + 4:nw + 7], X[321, :3]) and out[nw + 7] == 0
    assert hip.ya_slab_pack(dX.p, nw, n, payload.p, ws.p, None, 0, None, 0, limit, lag, None, 0, 0, 1, None, None) == 0
    out = payload.get(np.float32, nw + 8)
    assert out[nw + 2] == 0 and out[nw + 3] == 1 and (out[nw + 4:nw + 7] == 0).all()   # the host's error vote alone


def test_async_read_and_shader_clock(hip4):
    hip = hip4
    reader = C.c_void_p()
    assert hip.ya_async_read_create(8, C.byref(reader)) == 0
    src = Dev(hip, np.array([3.5, -1.25], np.float32))
    assert hip.ya_async_read_begin(reader, src.p, None) == 0
    got = np.zeros(2, np.float32)
    assert hip.ya_async_read_end(reader, got.ctypes.data) == 0 and list(got) == [3.5, -1.25]
    assert hip.ya_async_read_end(reader, got.ctypes.data) != 0, "nothing pending: refused"
    hip.ya_async_read_destroy(reader)
    mhz = C.c_double()
    assert hip.ya_shader_clock_mhz(300.0, C.byref(mhz)) == 0
    assert 300.0 < mhz.value < 4000.0, mhz.value


def test_loopback_communicators(hip4):
    """ya_comm_create_loopback: three communicators, a host thread each: the sum in rank order on every rank, a
    neighbour exchange with a size per message, several rounds (slots and events are reused)."""
    import threading
    hip = hip4
    world = 3
    handles = (C.c_void_p * world)()
    assert hip.ya_comm_create_loopback(world, handles) == 0
    comms = [C.c_void_p(h) for h in handles]
    rng = np.random.default_rng(2)
    contributions = [rng.random((6, 11)).astype(np.float32) for _ in range(world)]
    messages = {(r, d): rng.random(100 + 10 * r + d).astype(np.float32) for r in range(world) for d in (0, 1)}
    results, errors = [None] * world, []

    def work(r):
        try:
            sums = []
            buf = Dev(hip, nbytes=44)
            send = [Dev(hip, messages[(r, d)]) for d in (0, 1)]
            recv = [Dev(hip, nbytes=4 * 200) for _ in (0, 1)]
            for k in range(6):
                assert hip.ya_memcpy_h2d(buf.p, contributions[r][k].ctypes.data, 44) == 0
                assert hip.ya_comm_allreduce_sum(comms[r], buf.p, 11, None) == 0
                if k == 2:
                    n_lo = messages[(r - 1, 1)].nbytes if r > 0 else 0
                    n_hi = messages[(r + 1, 0)].nbytes if r + 1 < world else 0
                    assert hip.ya_comm_exchange_v(comms[r], send[0].p, messages[(r, 0)].nbytes, recv[0].p, n_lo,
                                                  send[1].p, messages[(r, 1)].nbytes, recv[1].p, n_hi, None) == 0
                assert hip.ya_device_synchronize() == 0
                sums.append(buf.get(np.float32, 11))
            got_lo = recv[0].get(np.float32, 200)[:len(messages[(r - 1, 1)])] if r > 0 else None
            got_hi = recv[1].get(np.float32, 200)[:len(messages[(r + 1, 0)])] if r + 1 < world else None
            results[r] = (sums, got_lo, got_hi)
        except Exception as err:      # pragma: no cover
            errors.append(err)

    threads = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not errors, errors
    for k in range(6):
        want = contributions[0][k].copy()
        for r in range(1, world):
            want = want + contributions[r][k]
        for r in range(world):
            assert np.array_equal(results[r][0][k].view(np.uint32), want.view(np.uint32)), (k, r)
    for r in range(world):
        if r > 0:
            assert np.array_equal(results[r][1], messages[(r - 1, 1)])
        if r + 1 < world:
            assert np.array_equal(results[r][2], messages[(r + 1, 0)])
    for c in comms:
        assert hip.ya_comm_destroy(c) == 0
