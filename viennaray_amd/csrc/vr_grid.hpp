// vr_grid.hpp — closest hit through the CELL GRID (gfx950): the per-lane strategy for rays that
// have lost their coherence (bounced rays in a structured scene).
//
// The per-lane BVH walk (vr_device.hpp: bvh_walk_lanes) is a chain of ~40 DEPENDENT 16-byte gathers
// per trace segment, each a separate cache line per lane; DESIGN.md 7 measures that chain — L1
// tag throughput and exposed L2 latency — as what bounds the bounce-heavy workloads, not
// instruction issue.  The cell grid replaces the chain by
//
//   * an occupancy map that lives in LDS: the scene box is cut into cubic cells of edge H, 4x4x4
//     cells form a BRICK, and one 64-bit word per brick says which of its cells hold geometry.
//     A ray steps through the grid with integer plane arithmetic (3-D DDA); an empty brick is
//     crossed in one step.  Empty space — most of what a ray crosses — costs LDS reads only.
//   * per occupied cell ONE header word and then a CONTIGUOUS run of primitive records (copies of
//     the records of every primitive whose padded box meets the cell; 288 GB of HBM pay for the
//     duplication): independent loads from a few consecutive cache lines instead of a pointer
//     chase.
//
// The closest-hit rule (min t, ties -> lower original id) does not depend on the order in which
// candidates are found, so the result is the BVH walk's bit for bit.  Conservativeness: a
// primitive is registered in every cell its box — padded by p.gridPad, far more than the rounding
// of the DDA's plane arithmetic — overlaps, so a hit point always lies in a visited cell that
// lists its primitive; the walk of a lane ends when the next cell begins behind its closest hit.
#pragma once
#include "vr_device.hpp"

namespace vr {

// cursor of a lane's grid walk (one register, resumable like the BVH cursor):
//   0          fresh: the segment has not entered the grid yet
//   VR_END     finished
//   otherwise  VR_GRID_LIVE | tested << 30 | iz << 20 | iy << 10 | ix   (the cell the lane stands in)
constexpr unsigned VR_GRID_LIVE = 0x80000000u;
constexpr unsigned VR_GRID_TESTED = 0x40000000u;

__device__ __forceinline__ bool grid_walking(unsigned cur) { return cur != VR_END; }

// closest-hit rule without the original id at hand: records of the cell lists carry the leaf
// position; on an exact tie between two different primitives (a measure-zero event) the ids are
// fetched from the primitive array
template <int GEO>
__device__ __forceinline__ void grid_hit_update(const TraceParams &p, HitRec &h, bool ok, float t, unsigned pos) {
  if (ok) {
    bool better = t < h.t;
    if (t == h.t && h.geom == 1 && pos != h.pos) {
      const float4 *__restrict__ prims = reinterpret_cast<const float4 *>(p.prims);
      const unsigned a = __float_as_uint(GEO == 0 ? prims[2 * (size_t)pos + 1].w : prims[4 * (size_t)pos].w);
      const unsigned b = __float_as_uint(GEO == 0 ? prims[2 * (size_t)h.pos + 1].w : prims[4 * (size_t)h.pos].w);
      better = a < b;
    }
    if (better) {
      h.t = t;
      h.geom = 1;
      h.pos = pos;
    }
  }
}

// `brickS`: the occupancy words in LDS.  `cur`: the lane's cursor (see above).  Runs while at least
// `minLanes` lanes of the wave are still walking, like bvh_walk_lanes.  h.prim is NOT maintained for
// geometry hits (the trace kernel works with h.pos; grid_fix_prim restores it for the debug entry).
template <int GEO>
__device__ __forceinline__ void grid_walk_lanes(const TraceParams &p, const unsigned long long *brickS, bool part,
                                                const V3 &o, const V3 &d, float tnear, HitRec &h, unsigned &cur,
                                                unsigned minLanes VR_DIAG_ARGS) {
  const float4 *__restrict__ recs = reinterpret_cast<const float4 *>(p.cellRecs);
  const float H = p.gridH, invH = p.gridInvH;
  const int nx = (int)p.gridDim[0], ny = (int)p.gridDim[1], nz = (int)p.gridDim[2];
  const int bnx = (int)p.brickDim[0], bny = (int)p.brickDim[1];
  const V3 inv = safe_inverse(d);
  // t of grid plane k on axis a: fma(k, invs_a, oiw_a)
  const V3 invs = V3{inv.x * H, inv.y * H, inv.z * H};
  const V3 oiw = V3{(p.gridLo[0] - o.x) * inv.x, (p.gridLo[1] - o.y) * inv.y, (p.gridLo[2] - o.z) * inv.z};
  // step direction per axis = the SIGN BIT of d (safe_inverse keeps it: -0 steps down with inv = -1e30)
  const bool sx = (__float_as_uint(d.x) >> 31) == 0u, sy = (__float_as_uint(d.y) >> 31) == 0u,
             sz = (__float_as_uint(d.z) >> 31) == 0u;
  const float invDD = 1.0f / vdot(d, d);
  if (!part)
    cur = VR_END;
  int ix = 0, iy = 0, iz = 0;
  bool tested = false;
  if (cur == 0u) {
    // enter: clip the ray to the grid's box
    const float ax0 = oiw.x, ax1 = __builtin_fmaf((float)nx, invs.x, oiw.x);
    const float ay0 = oiw.y, ay1 = __builtin_fmaf((float)ny, invs.y, oiw.y);
    const float az0 = oiw.z, az1 = __builtin_fmaf((float)nz, invs.z, oiw.z);
    const float tIn = fmaxf(fmaxf(fminf(ax0, ax1), fminf(ay0, ay1)), fmaxf(fminf(az0, az1), 0.f));
    const float tOut = fminf(fminf(fmaxf(ax0, ax1), fmaxf(ay0, ay1)), fmaxf(az0, az1));
    if (tIn <= tOut) {
      const int qx = (int)floorf((__builtin_fmaf(d.x, tIn, o.x) - p.gridLo[0]) * invH);
      const int qy = (int)floorf((__builtin_fmaf(d.y, tIn, o.y) - p.gridLo[1]) * invH);
      const int qz = (int)floorf((__builtin_fmaf(d.z, tIn, o.z) - p.gridLo[2]) * invH);
      ix = min(max(qx, 0), nx - 1);
      iy = min(max(qy, 0), ny - 1);
      iz = min(max(qz, 0), nz - 1);
      cur = VR_GRID_LIVE;
    } else {
      cur = VR_END;
    }
  } else if (cur != VR_END) {
    ix = (int)(cur & 1023u);
    iy = (int)((cur >> 10) & 1023u);
    iz = (int)((cur >> 20) & 1023u);
    tested = (cur & VR_GRID_TESTED) != 0u;
  }
  bool alive = cur != VR_END;
  bool havePend = false;
  unsigned pendHdr = 0u;
  for (;;) {
    for (;;) {
      const unsigned long long sm = ballot64(alive && !havePend);
      if (!sm)
        break;
      const unsigned long long km = ballot64(havePend);
      if (100u * (unsigned)__popcll(km) >= p.walkPark * (unsigned)__popcll(km | sm))
        break;
#pragma unroll
      for (int rep = 0; rep < 2; ++rep) {
        const bool search = alive && !havePend;
#ifdef VR_DIAG
        if (search) {
          DIAG(1);
        }
#endif
        const int bidx = ((iz >> 2) * bny + (iy >> 2)) * bnx + (ix >> 2);
        const unsigned long long mask = brickS[search ? bidx : 0];
        const unsigned bit = (unsigned)(ix & 3) | ((unsigned)(iy & 3) << 2) | ((unsigned)(iz & 3) << 4);
        const bool occ = ((mask >> bit) & 1ull) != 0ull;
        const bool empty = mask == 0ull;
        const bool take = search && occ && !tested;
        if (take)
          pendHdr = p.cellHdr[((size_t)iz * ny + iy) * nx + ix]; // (exec-masked; in flight while the others search on)
        havePend = havePend || take;
        tested = tested || take;
        const bool adv = search && !take;
        // next plane per axis: the cell's, or — in an empty brick — the brick's
        const int lox = empty ? (ix & ~3) : ix, loy = empty ? (iy & ~3) : iy, loz = empty ? (iz & ~3) : iz;
        const int hix = empty ? lox + 3 : ix, hiy = empty ? loy + 3 : iy, hiz = empty ? loz + 3 : iz;
        const int px = sx ? hix + 1 : lox, py = sy ? hiy + 1 : loy, pz = sz ? hiz + 1 : loz;
        const float tx = __builtin_fmaf((float)px, invs.x, oiw.x);
        const float ty = __builtin_fmaf((float)py, invs.y, oiw.y);
        const float tz = __builtin_fmaf((float)pz, invs.z, oiw.z);
        const float tN = fminf(fminf(tx, ty), tz);
        const bool ax = tx <= ty && tx <= tz;
        const bool ay = !ax && ty <= tz;
        const bool az = !ax && !ay;
        // the axes that do not cross their plane: cell from the position at tN, never backwards,
        // never out of the brick just crossed
        const int qx = (int)floorf((__builtin_fmaf(d.x, tN, o.x) - p.gridLo[0]) * invH);
        const int qy = (int)floorf((__builtin_fmaf(d.y, tN, o.y) - p.gridLo[1]) * invH);
        const int qz = (int)floorf((__builtin_fmaf(d.z, tN, o.z) - p.gridLo[2]) * invH);
        const int cx = sx ? min(max(qx, ix), hix) : max(min(qx, ix), lox);
        const int cy = sy ? min(max(qy, iy), hiy) : max(min(qy, iy), loy);
        const int cz = sz ? min(max(qz, iz), hiz) : max(min(qz, iz), loz);
        const int mx = ax ? (sx ? px : px - 1) : cx;
        const int my = ay ? (sy ? py : py - 1) : cy;
        const int mz = az ? (sz ? pz : pz - 1) : cz;
        const bool out = mx < 0 || mx >= nx || my < 0 || my >= ny || mz < 0 || mz >= nz || tN * 0.99999f > h.t;
        ix = adv ? mx : ix;
        iy = adv ? my : iy;
        iz = adv ? mz : iz;
        tested = adv ? false : tested;
        alive = alive && !(adv && out);
      }
    }
    if (ballot64(havePend)) {
      const unsigned first = pendHdr >> 6;
      const unsigned cnt = havePend ? (pendHdr & 63u) : 0u;
      for (unsigned i = 0; ballot64(i < cnt); ++i) {
        const bool on = i < cnt;
#ifdef VR_DIAG
        if (on) {
          DIAG(2);
        }
#endif
        const unsigned q = on ? first + i : 0u;
        float t;
        if (GEO == 0) {
          const float4 c4 = recs[2 * (size_t)q];
          if (on && disc_may_hit(o, d, invDD, c4)) {
            const float4 n4 = recs[2 * (size_t)q + 1];
            const bool ok = hit_disc(o, d, tnear, c4, mk(n4.x, n4.y, n4.z), t);
            grid_hit_update<GEO>(p, h, ok, t, __float_as_uint(n4.w));
          }
        } else {
          const float4 a = recs[4 * (size_t)q], b = recs[4 * (size_t)q + 1], c = recs[4 * (size_t)q + 2],
                       e = recs[4 * (size_t)q + 3];
          const bool ok =
              hit_tri(o, d, tnear, mk(a.x, a.y, a.z), mk(b.x, b.y, b.z), mk(c.x, c.y, c.z), mk(e.x, e.y, e.z), t);
          grid_hit_update<GEO>(p, h, on && ok, t, __float_as_uint(a.w));
        }
      }
      havePend = false;
    }
    if ((unsigned)__popcll(ballot64(alive)) < minLanes)
      break;
  }
  cur = alive ? (VR_GRID_LIVE | (tested ? VR_GRID_TESTED : 0u) | ((unsigned)iz << 20) | ((unsigned)iy << 10) | (unsigned)ix)
              : VR_END;
}

// h.prim of a geometry hit found by the grid walk (diagnostic entry points compare it)
template <int GEO> __device__ __forceinline__ void grid_fix_prim(const TraceParams &p, HitRec &h) {
  if (h.geom == 1) {
    const float4 *__restrict__ prims = reinterpret_cast<const float4 *>(p.prims);
    h.prim = __float_as_uint(GEO == 0 ? prims[2 * (size_t)h.pos + 1].w : prims[4 * (size_t)h.pos].w);
  }
}

} // namespace vr
