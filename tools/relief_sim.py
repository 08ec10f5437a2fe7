#!/usr/bin/env python3
"""CPU estimate (numpy, no GPU) of what a wave's packet query has to test on a sheet with relief, under different ways of
clipping its rays and of binning them: how many disks meet the wave's query box Q.  Used to decide what to build for
"flat with relief" scenes before any kernel was written.

usage: tools/relief_sim.py [amp=0.5] [wave=4.0] [waves=400]
"""
import sys
import numpy as np

amp = float(sys.argv[1]) if len(sys.argv) > 1 else 0.5
wl = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
NW = int(sys.argv[3]) if len(sys.argv) > 3 else 400
rng = np.random.default_rng(1)
n = 1000
r = 0.8660254 * (1 + 1e-5)
ax = np.arange(n) - (n - 1) / 2.0


def h(x, y):
    return amp * np.sin(x / wl) * np.cos(y / wl)


X, Y = np.meshgrid(ax, ax, indexing="ij")
Z = h(X, Y)
nx_ = -amp / wl * np.cos(X / wl) * np.cos(Y / wl)
ny_ = amp / wl * np.sin(X / wl) * np.sin(Y / wl)
nn = np.sqrt(nx_ ** 2 + ny_ ** 2 + 1)
NZ = 1 / nn
# disk boxes
ex = r * np.sqrt(1 - (nx_ / nn) ** 2)
ey = r * np.sqrt(1 - (ny_ / nn) ** 2)
ez = r * np.sqrt(1 - NZ ** 2)
zlo_g, zhi_g = (Z - ez).min(), (Z + ez).max()
print(f"scene slab [{zlo_g:.3f}, {zhi_g:.3f}] thickness {zhi_g - zlo_g:.3f}")

# fine min/max tiles of side T (grid cells): every disk whose box meets the tile
def tiles(T):
    m = int(np.ceil(n / T))
    lo = np.full((m, m), 1e9)
    hi = np.full((m, m), -1e9)
    x0 = ax[0] - 0.5
    for dx in (-1, 0, 1):
        for dy in (-1, 0, 1):
            # a disk reaches at most r < 1 cell: tiles of its centre +- 1 cell cover it (conservative)
            ix = np.clip(((X + dx * r - x0) / T).astype(int), 0, m - 1)
            iy = np.clip(((Y + dy * r - x0) / T).astype(int), 0, m - 1)
            np.minimum.at(lo, (ix, iy), Z - ez)
            np.maximum.at(hi, (ix, iy), Z + ez)
    return lo, hi, x0, T


TL = tiles(1.0)


def count_disks(qlx, qhx, qly, qhy, qlz, qhz):
    i0 = max(0, int(np.floor(qlx - r - ax[0])))
    i1 = min(n - 1, int(np.ceil(qhx + r - ax[0])))
    j0 = max(0, int(np.floor(qly - r - ax[0])))
    j1 = min(n - 1, int(np.ceil(qhy + r - ax[0])))
    if i1 < i0 or j1 < j0:
        return 0
    xs, ys, zs = X[i0:i1 + 1, j0:j1 + 1], Y[i0:i1 + 1, j0:j1 + 1], Z[i0:i1 + 1, j0:j1 + 1]
    ok = (xs - r <= qhx) & (xs + r >= qlx) & (ys - r <= qhy) & (ys + r >= qly) & (zs - r <= qhz) & (zs + r >= qlz)
    return int(ok.sum())


def sample_dirs(k):
    u1, u2 = rng.random(k), rng.random(k)
    ct = np.sqrt(u2)
    st = np.sqrt(1 - ct * ct)
    ph = 2 * np.pi * u1
    return np.stack([st * np.cos(ph), st * np.sin(ph), -ct], -1)


def dda_clip(p0, d, lo, hi, x0, T, maxsteps=64):
    """per ray: t-range [tA, tB] (t = 0 at p0 on the slab's top) covering every tile in which the ray's height range
    overlaps the tile's [lo, hi]; steps = tiles visited"""
    tA, tB, steps = np.inf, -np.inf, 0
    tend = (zlo_g - p0[2]) / d[2]
    t = 0.0
    m = lo.shape[0]
    while t < tend and steps < maxsteps:
        p = p0 + d * (t + 1e-9)
        ix, iy = int(np.floor((p[0] - x0) / T)), int(np.floor((p[1] - x0) / T))
        # exit of this tile
        tx = ((x0 + (ix + (d[0] > 0)) * T) - p0[0]) / d[0] if d[0] != 0 else np.inf
        ty = ((x0 + (iy + (d[1] > 0)) * T) - p0[1]) / d[1] if d[1] != 0 else np.inf
        tn = min(tx, ty, tend)
        steps += 1
        if 0 <= ix < m and 0 <= iy < m:
            z0, z1 = p0[2] + d[2] * t, p0[2] + d[2] * tn
            if z1 <= hi[ix, iy] and z0 >= lo[ix, iy]:
                # clip inside the tile to the tile's slab
                ta = max(t, (hi[ix, iy] - p0[2]) / d[2])
                tb = min(tn, (lo[ix, iy] - p0[2]) / d[2])
                tA, tB = min(tA, ta), max(tB, tb)
        t = tn
    return tA, tB, steps


cell = 1000.0 / np.sqrt(1e8 / 40)   # side of a sort bin's cell
res = {}
for scheme in ("global", "global_tan2", "dda_plane_tan2", "dda_pred_tan2", "dda_pred_tan3", "dda_pred_all"):
    cnts, steps_all = [], []
    tmax = {"global": 1e9, "global_tan2": 2, "dda_plane_tan2": 2, "dda_pred_tan2": 2, "dda_pred_tan3": 3, "dda_pred_all": 1e9}[scheme]
    for w in range(NW):
        cx, cy = rng.uniform(-400, 400, 2)
        k = 64
        d = sample_dirs(4 * k)
        tan = np.sqrt(d[:, 0] ** 2 + d[:, 1] ** 2) / -d[:, 2]
        d = d[tan <= tmax][:k]
        k = len(d)
        # where the ray is sorted to: 1.6 adjacent cells
        px = cx + rng.uniform(0, 1.6 * cell, k)
        py = cy + rng.uniform(0, cell, k)
        if "pred" in scheme:
            # sorted by the predicted hit: the ray passes through (px, py, h(px, py))
            pz = h(px, py)
        else:
            pz = np.zeros(k)  # sort plane z = 0
        # entry into the global slab
        t0 = (zhi_g - pz) / d[:, 2]
        p0 = np.stack([px, py, pz], -1) + d * t0[:, None]
        t1 = (zlo_g - zhi_g) / d[:, 2]
        if scheme.startswith("global"):
            a, b = p0, p0 + d * t1[:, None]
            valid = np.ones(k, bool)
        else:
            a, b, valid = np.zeros_like(p0), np.zeros_like(p0), np.zeros(k, bool)
            for i in range(k):
                tA, tB, st = dda_clip(p0[i], d[i], *TL)
                steps_all.append(st)
                if tB >= tA:
                    a[i], b[i], valid[i] = p0[i] + d[i] * tA, p0[i] + d[i] * tB, True
        if not valid.any():
            cnts.append(0)
            continue
        lo3 = np.minimum(a[valid], b[valid]).min(0)
        hi3 = np.maximum(a[valid], b[valid]).max(0)
        cnts.append(count_disks(lo3[0], hi3[0], lo3[1], hi3[1], lo3[2], hi3[2]))
    c = np.array(cnts)
    msg = f"{scheme:16s} candidates: mean {c.mean():6.1f} median {np.median(c):5.0f} p90 {np.percentile(c, 90):5.0f}  <=24: {np.mean(c <= 24) * 100:5.1f} %  <=40: {np.mean(c <= 40) * 100:5.1f} %"
    if steps_all:
        s = np.array(steps_all)
        msg += f"   dda steps mean {s.mean():.1f} p99 {np.percentile(s, 99):.0f} max {s.max()}"
    print(msg)
