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


def lines_to_triangles(nodes, lines, grid_delta):
    """LineMesh -> TriangleMesh (rayMesh.hpp:27-80,133-175): zero-length lines are dropped,
    every remaining line becomes a strip of two triangles of height grid_delta
    (node 2i = node i at z = +grid_delta/2, node 2i+1 at z = -grid_delta/2).
    Returns (vertices float32[2V,3], triangles uint32[2L,3], kept line indices)."""
    nodes = np.ascontiguousarray(nodes, dtype=np.float32).reshape(-1, 3)
    lines = np.ascontiguousarray(lines, dtype=np.uint32).reshape(-1, 2)
    d = nodes[lines[:, 1]] - nodes[lines[:, 0]]
    length = np.sqrt((d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1] + d[:, 2] * d[:, 2]).astype(np.float32))
    keep = np.nonzero(length > np.float32(1e-6))[0]
    w2 = np.float32(grid_delta) * np.float32(0.5)
    verts = np.empty((2 * nodes.shape[0], 3), dtype=np.float32)
    verts[0::2, :2] = nodes[:, :2]
    verts[1::2, :2] = nodes[:, :2]
    verts[0::2, 2] = w2
    verts[1::2, 2] = -w2
    p0 = lines[keep, 0] * 2
    p1 = lines[keep, 1] * 2
    tris = np.empty((2 * keep.size, 3), dtype=np.uint32)
    tris[0::2] = np.stack([p0, p1, p0 + 1], 1)
    tris[1::2] = np.stack([p0 + 1, p1, p1 + 1], 1)
    return verts, tris, keep


def read_line_mesh(path):
    """2-D mesh file as the reference's reader sees it (rayUtil.hpp:374-411): elements the
    file declares but does not hold stay (0, 0)."""
    declared = None
    with open(path) as fh:
        for line in fh:
            t = line.split()
            if t and t[0] == "n_elements":
                declared = int(t[1])
                break
    gd, nodes, lines = read_mesh(path, 2)
    if declared is not None and lines.shape[0] < declared:
        lines = np.concatenate([lines, np.zeros((declared - lines.shape[0], 2), np.uint32)])
    return gd, nodes, lines
