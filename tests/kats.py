"""The reference's hot-path known-answer tests, restated once and run against
either backend (CPU oracle or HIP engine) through the same harness ABI.

Sources: reference tests/test_solvers.cu (10 cases, Gabriel excluded: out of
scope) and tests/test_links.cu (2 cases), with the 3-argument gen_forces
signature the headers actually have.  Tolerance is the
reference's own `isclose`: |a - b| <= 1e-6 + 1e-2 |b| (tests/minunit.cuh:37),
exact where the reference is exact.
"""
import math

import numpy as np

from yalla_amd.solution import Solution


def isclose(a, b):
    return abs(a - b) <= 1e-6 + 1e-2 * abs(b)


def center_of_mass(cells):  # tests/minunit.cuh:43-53
    n = cells.h_n
    return (cells.h_X[:n, :3].astype(np.float32) / np.float32(n)).sum(axis=0)


def kat_oscillation(lib):  # test_solvers.cu:8-39
    with Solution("oscillator_tile", 2, lib=lib) as osc:
        osc.h_X[:] = 0
        osc.h_X[0, 3] = 1
        osc.h_X[1, 3] = 0
        osc.copy_to_device()
        n_steps = 100
        for _ in range(n_steps):
            osc.take_step(2 * math.pi / n_steps)
            osc.copy_to_host()
            assert isclose(osc.h_X[0, 3] ** 2 + osc.h_X[1, 3] ** 2, 1), "Oscillator off circle"
        assert isclose(osc.h_X[0, 3], 1), "Oscillator final cosine"


def kat_tetrahedron(lib, solver, seed=7):  # test_solvers.cu:55-98
    with Solution(f"clipped_{solver}", 4, lib=lib) as pts:
        pts.random_sphere(0.5, seed)
        pts.copy_to_host()
        com_i = center_of_mass(pts)
        pts.take_step(0.1, 500)
        pts.copy_to_host()
        for i in range(1, 4):
            r = pts.h_X[0, :3] - pts.h_X[i, :3]
            assert isclose(float(np.sqrt((r * r).sum())), 0.5), "Spring not relaxed"
        com_f = center_of_mass(pts)
        for k in range(3):
            assert isclose(com_i[k], com_f[k]), "Momentum in tetrahedron"


def kat_compare_methods(lib, seed=11):  # test_solvers.cu:100-125
    n = 50
    with Solution("clipped_tile", n, lib=lib) as tile, Solution("clipped_grid", n, lib=lib) as grid:
        tile.random_sphere(0.733333, seed)
        grid.h_X[:] = tile.h_X
        grid.copy_to_device()
        tile.take_step(0.1, 2)
        grid.take_step(0.1, 2)
        a, b = tile.positions(), grid.positions()
        for i in range(n):
            for k in range(3):
                assert isclose(a[i, k], b[i, k]), "Methods disagree"


def kat_generic_forces(lib):  # test_solvers.cu:146-183
    for model in ("push_tile", "clipped_push_grid"):
        with Solution(model, 2, lib=lib) as pts:
            pts.h_X[0] = (0, 0, 10)
            pts.h_X[1] = (0, 0, 0)
            pts.copy_to_device()
            com_i = center_of_mass(pts)
            pts.take_step(1.0)
            pts.copy_to_host()
            com_f = center_of_mass(pts)
            for k in range(3):
                assert isclose(com_i[k], com_f[k]), f"Momentum in {model} generic force"
            assert isclose(pts.h_X[1, 0], 0.5), f"{model} generic force failed in x"
            assert isclose(pts.h_X[1, 1], 0), f"{model} generic force failed in y"
            assert isclose(pts.h_X[1, 2], 0), f"{model} generic force failed in z"


def kat_friction(lib):  # test_solvers.cu:186-225
    for solver in ("tile", "grid"):
        for model, expected in ((f"push_background_{solver}", 1.0), (f"push_{solver}", 0.75)):
            with Solution(model, 2, lib=lib) as pts:
                pts.h_X[0] = (0, 0, 0)
                pts.h_X[1] = (0.5, 0, 0)
                pts.copy_to_device()
                pts.take_step(0.05, 10)
                pts.copy_to_host()
                assert isclose(pts.h_X[1, 0] - pts.h_X[0, 0], expected), f"{model} friction"


def kat_fix_point(lib, seed=13):  # test_solvers.cu:228-244
    with Solution("clipped_tile", 100, lib=lib) as tile:
        tile.random_sphere(0.733333, seed)
        fix_point = 13
        tile.h_X[fix_point] = 0
        tile.copy_to_device()
        tile.set_fixed(fix_point)
        tile.take_step(0.1)
        tile.copy_to_host()
        for k in range(3):
            assert isclose(tile.h_X[fix_point, k], 0), "Fixed point moved"


def lattice(n_x=7, n_y=7, n_z=7):
    k, j, i = np.meshgrid(np.arange(n_x), np.arange(n_y), np.arange(n_z), indexing="ij")
    X = np.zeros((n_x * n_y * n_z, 3), np.float32)
    idx = (n_x * n_y * i + n_x * j + k).ravel()
    X[idx, 0] = k.ravel() + 0.5
    X[idx, 1] = j.ravel() + 0.5
    X[idx, 2] = i.ravel() + 0.5
    return X


def kat_grid_spacing(lib):  # test_solvers.cu:247-315 -- exact integers
    n_x = n_y = n_z = 7
    gs = 70
    with Solution("clipped_grid", n_x * n_y * n_z, lib=lib) as pts:
        pts.h_X[:] = lattice(n_x, n_y, n_z)
        pts.copy_to_device()
        origin = gs ** 3 // 2 + gs ** 2 // 2 + gs // 2
        tz, ty, tx = np.meshgrid(np.arange(n_z), np.arange(n_y), np.arange(n_x), indexing="ij")
        i = (tx + n_x * ty + n_x * n_y * tz).ravel()

        cube_id, point_id, start, end = pts.build_grid(gs, 1.0)  # single_grid
        expected = (origin + tx + gs * ty + gs * gs * tz).ravel()
        assert (start[expected] == end[expected]).all(), "one point per cube"
        assert (cube_id[i] == expected).all(), "cube ids"
        assert (point_id == np.arange(len(i))).all()

        cube_id, point_id, start, end = pts.build_grid(gs, 2.0)  # double_grid
        expected = (origin + tx // 2 + gs * (ty // 2) + gs * gs * (tz // 2)).ravel()
        for cell, cube in zip(i, expected):
            members = point_id[start[cube]: end[cube] + 1]
            assert cell in members, "cell not in expected cube"
        return cube_id, point_id, start, end


def kat_cube_size(lib):  # test_solvers.cu:318-336 -- exact ==
    with Solution("clipped_grid", 2, lib=lib) as pts:
        pts.h_X[0] = 0
        pts.h_X[1] = (0.75, 0, 0)
        pts.copy_to_device()
        pts.cube_size = 0.5
        pts.take_step(0.1)
        pts.copy_to_host()
        assert pts.h_X[0, 0] == 0, "Cell outside cube moved"
        pts.cube_size = 1
        pts.take_step(0.1)
        pts.copy_to_host()
        assert pts.h_X[0, 0] != 0, "Cell inside cube did not move"


def kat_square_of_four(lib):  # test_links.cu:15-50
    with Solution("links_tile", 4, lib=lib) as pts:
        pts.h_X[:] = [(1, 1, 0), (1, -1, 0), (-1, -1, 0), (-1, 1, 0)]
        pts.copy_to_device()
        pts.set_links([(0, 1), (1, 2), (2, 3), (3, 0)])
        com_i = center_of_mass(pts)
        pts.take_step(0.1, 500)
        pts.copy_to_host()
        com_f = center_of_mass(pts)
        for k in range(3):
            assert isclose(com_i[k], com_f[k]), "Momentum in square"
        assert isclose(pts.h_X[0, 0], pts.h_X[1, 0]), "Not close in x"
        assert isclose(pts.h_X[1, 1], pts.h_X[2, 1]), "Not close in y"
        assert isclose(pts.h_X[2, 2], pts.h_X[3, 2]), "Not close in z"


def kat_custom_force(lib):  # test_links.cu:61-93
    dt, strength = 0.1, 0.2
    X0 = [(1, 1, 0, 1), (1, -1, 0, -1)]
    with Solution("links4_tile", 2, lib=lib) as a:
        a.h_X[:] = X0
        a.copy_to_device()
        a.set_links([(0, 1)], strength)
        a.take_step(dt)
        mid = a.positions()
    with Solution("links4_custom_tile", 2, lib=lib) as b:
        b.h_X[:] = mid
        b.copy_to_device()
        b.set_links([(0, 1)], strength)
        b.take_step(dt)
        X = b.positions()
    assert isclose(X[0, 0] - X[1, 0], 0), "Wrong x"
    assert isclose(X[0, 1] - X[1, 1], 2 - 2 * dt * strength), "Wrong y"
    assert isclose(X[0, 2] - X[1, 2], 0), "Wrong z"
    assert isclose(X[0, 3] - X[1, 3], 2 - 2 * dt), "Wrong w"


ALL = [
    ("oscillation", kat_oscillation),
    ("tile_tetrahedron", lambda lib: kat_tetrahedron(lib, "tile")),
    ("grid_tetrahedron", lambda lib: kat_tetrahedron(lib, "grid")),
    ("compare_methods", kat_compare_methods),
    ("generic_forces", kat_generic_forces),
    ("friction", kat_friction),
    ("fix_point", kat_fix_point),
    ("grid_spacing", kat_grid_spacing),
    ("cube_size", kat_cube_size),
    ("square_of_four", kat_square_of_four),
    ("custom_force", kat_custom_force),
]
