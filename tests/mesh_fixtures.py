"""Input files for the mesh tests, GENERATED (nothing here is a file of the reference's):

  write_torus      a closed triangulated torus, ring radius 1, tube radius 0.5, about the z axis -- the
                   surface the reference's tests/test_mesh.cu asserts on analytically (extent +-1.5 / +-0.5,
                   exclusion against the distance from the ring, growth along the normals), as legacy ASCII
                   VTK the way its `tests/torus.vtk` is laid out: several points per line, `POLYGONS m 4m`
  write_sphere_ic  a ball of cells for examples/intercalation_w_gradient.cu, which loads its initial
                   condition from `examples/sphere_ic.vtk`: positions, `NORMALS polarity`, `SCALARS cell_type`
                   (0 inside, 1 in the outermost shell with the outward normal as polarity), in the layout
                   Vtk_output writes
"""
import numpy as np


def torus(n_ring=64, n_tube=32, ring=1.0, tube=0.5):
    """Vertices [n_ring * n_tube, 3] on the torus and triangles [2 * n_ring * n_tube, 3] with outward normals."""
    u = 2 * np.pi * np.arange(n_ring) / n_ring
    v = 2 * np.pi * np.arange(n_tube) / n_tube
    uu, vv = np.meshgrid(u, v, indexing="ij")
    rho = ring + tube * np.cos(vv)
    verts = np.stack([rho * np.cos(uu), rho * np.sin(uu), tube * np.sin(vv)], axis=-1).reshape(-1, 3)
    tris = []
    for a in range(n_ring):
        for b in range(n_tube):
            p00 = a * n_tube + b
            p10 = ((a + 1) % n_ring) * n_tube + b
            p01 = a * n_tube + (b + 1) % n_tube
            p11 = ((a + 1) % n_ring) * n_tube + (b + 1) % n_tube
            tris += [(p00, p10, p11), (p00, p11, p01)]
    return verts.astype(np.float32), np.array(tris, dtype=np.int32)


def write_mesh(path, verts, tris, per_line=3):
    with open(path, "w") as f:
        f.write("# vtk DataFile Version 4.0\nvtk output\nASCII\nDATASET POLYDATA\n")
        f.write(f"POINTS {len(verts)} float\n")
        for k in range(0, len(verts), per_line):
            f.write(" ".join(f"{c:.6g}" for p in verts[k:k + per_line] for c in p) + " \n")
        f.write(f"POLYGONS {len(tris)} {4 * len(tris)}\n")
        for t in tris:
            f.write(f"3 {t[0]} {t[1]} {t[2]} \n")


def write_torus(path, **kw):
    verts, tris = torus(**kw)
    write_mesh(path, verts, tris)
    return verts, tris


def write_sphere_ic(path, radius=6.0, spacing=0.75, shell=0.55):
    """Cells on a face-centred cubic lattice inside a ball; returns (positions, cell_type)."""
    a = spacing * np.sqrt(2)          # cubic cell edge for nearest-neighbour distance `spacing`
    k = int(np.ceil(radius / a)) + 1
    grid = np.arange(-k, k + 1)
    base = np.array([(0, 0, 0), (0.5, 0.5, 0), (0.5, 0, 0.5), (0, 0.5, 0.5)])
    cells = (np.stack(np.meshgrid(grid, grid, grid, indexing="ij"), -1).reshape(-1, 1, 3) + base).reshape(-1, 3) * a
    cells = cells + 0.013          # off the planes x, y, z == 0 the model branches on
    r = np.linalg.norm(cells, axis=1)
    cells, r = cells[r <= radius], r[r <= radius]
    cell_type = (r > radius - shell).astype(int)
    normals = np.where(cell_type[:, None] == 1, cells / r[:, None], 0.0)
    n = len(cells)
    with open(path, "w") as f:
        f.write("# vtk DataFile Version 3.0\nsphere_ic\nASCII\nDATASET POLYDATA\n")
        f.write(f"\nPOINTS {n} float\n")
        f.writelines(f"{p[0]:.6g} {p[1]:.6g} {p[2]:.6g}\n" for p in cells)
        f.write(f"\nVERTICES {n} {2 * n}\n")
        f.writelines(f"1 {i}\n" for i in range(n))
        f.write(f"\nPOINT_DATA {n}\nNORMALS polarity float\n")
        f.writelines(f"{p[0]:.6g} {p[1]:.6g} {p[2]:.6g}\n" for p in normals)
        f.write("SCALARS cell_type int\nLOOKUP_TABLE default\n")
        f.writelines(f"{t}\n" for t in cell_type)
    return cells, cell_type
