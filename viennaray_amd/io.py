"""Synthetic geometry generators and readers for the reference's `.dat` fixtures.

Formats follow the reference readers (`include/viennaray/rayUtil.hpp:353-411`);
`create_plane_grid` follows `createPlaneGrid` (`rayUtil.hpp:324-351`) including
its float accumulation, `plane_grid` is the integer-indexed variant the bench
uses (index = i*n + j, no accumulation drift).
"""
import numpy as np


def create_plane_grid(grid_delta, extent, direction=(0, 1, 2)):
    """rayUtil.hpp:324-351 (NumericType = float)."""
    f = np.float32
    grid_delta, extent = f(grid_delta), f(extent)
    d0, d1, d2 = direction
    point = np.array([-extent, -extent, -extent], dtype=f)
    normal = np.zeros(3, dtype=f)
    point[d2] = 0
    normal[d2] = 1
    pts = []
    while point[d0] <= extent:
        while point[d1] <= extent:
            pts.append(point.copy())
            point[d1] = f(point[d1] + grid_delta)
        point[d1] = -extent
        point[d0] = f(point[d0] + grid_delta)
    pts = np.array(pts, dtype=f).reshape(-1, 3)
    nrm = np.tile(normal, (pts.shape[0], 1))
    return pts, nrm


def plane_grid(n, grid_delta=1.0):
    """P(n): x=(i-(n-1)/2)d, y=(j-(n-1)/2)d, z=0, normal +z, index i*n+j."""
    idx = (np.arange(n, dtype=np.float64) - (n - 1) / 2.0) * grid_delta
    x, y = np.meshgrid(idx, idx, indexing="ij")
    pts = np.stack([x.ravel(), y.ravel(), np.zeros(n * n)], axis=1).astype(np.float32)
    nrm = np.zeros_like(pts)
    nrm[:, 2] = 1.0
    return pts, nrm


def read_grid(path):
    """Disk grid: `numPoints gridDelta`, then points, then normals."""
    with open(path) as fh:
        tok = fh.read().split()
    n = int(tok[0])
    grid_delta = float(tok[1])
    vals = np.array(tok[2:2 + 6 * n], dtype=np.float64).astype(np.float32)
    pts = vals[:3 * n].reshape(n, 3)
    nrm = vals[3 * n:6 * n].reshape(n, 3)
    return grid_delta, pts, nrm


def read_mesh(path, dim=3):
    """Mesh: grid_delta / n_nodes / n_elements, then `n x y z`, `e i j [k]`.
    Tolerates a short element count (lineMesh.dat declares 130, holds 129)."""
    grid_delta = None
    nodes, elems = [], []
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if not t:
                continue
            if t[0] == "grid_delta":
                grid_delta = float(t[1])
            elif t[0] == "n":
                nodes.append([float(v) for v in t[1:4]])
            elif t[0] == "e":
                elems.append([int(v) for v in t[1:1 + dim]])
    return (grid_delta, np.array(nodes, dtype=np.float32),
            np.array(elems, dtype=np.uint32))
