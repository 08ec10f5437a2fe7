// vr_host.cpp — host-side setup (see vr_host.hpp).  Reference citations are
// relative to /root/reference/include/viennaray/.
#include "vr_host.hpp"

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <thread>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

namespace vr {

namespace {
struct F3 {
  float x, y, z;
};
inline float dot3(const F3 &a, const F3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
inline F3 cross3(const F3 &a, const F3 &b) {
  return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline F3 sub3(const F3 &a, const F3 &b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline float norm3(const F3 &a) { return std::sqrt(dot3(a, a)); }
inline void normalize3(F3 &a) {
  float n = norm3(a);
  if (n <= 0.f)
    return;
  a.x /= n;
  a.y /= n;
  a.z /= n;
}
} // namespace

// ---------------------------------------------------------------------------
// geometry ingestion
// ---------------------------------------------------------------------------
// splits [0, n) over a few host threads (the per-geometry host loops run once per new point
// cloud, on 10^6+ points: bbox, disk areas, sort-plane histogram); fn(thread, begin, end)
template <class F> static void parallel_ranges(uint32_t n, F fn) {
  unsigned hw = std::thread::hardware_concurrency();
  unsigned nt = n < (1u << 16) ? 1u : std::min(16u, std::max(1u, hw));
  if (nt <= 1) {
    fn(0u, 0u, n);
    return;
  }
  std::vector<std::thread> th;
  const uint32_t chunk = (n + nt - 1) / nt;
  for (unsigned t = 0; t < nt; ++t) {
    const uint32_t b = std::min<uint64_t>((uint64_t)t * chunk, n), e = std::min<uint64_t>((uint64_t)(t + 1) * chunk, n);
    th.emplace_back([=] { fn(t, b, e); });
  }
  for (auto &x : th)
    x.join();
}
constexpr unsigned kMaxHostThreads = 16;

void host_set_disks(HostGeometry &g, const float *pts, const float *nrm, uint32_t n, float gridDelta, float radius,
                    int D) {
  g.D = D;
  g.geo = 0;
  g.numPrims = n;
  g.gridDelta = gridDelta;
  // rayTraceDisk.hpp:70 / rayUtil.hpp:99-101
  const double factor = 0.5 * (D == 3 ? 1.7320508 : 1.41421356237) * (1 + 1e-5);
  g.diskRadius = radius > 0.f ? radius : (float)(gridDelta * factor);
  g.disk4.resize((size_t)n * 4);
  g.normal3.resize((size_t)n * 3);
  g.points3.resize((size_t)n * 3); // (filled by the threads below: a serial copy of 12 MB was half of this function)
  for (int k = 0; k < D; ++k) {
    g.minC[k] = std::numeric_limits<float>::max();
    g.maxC[k] = std::numeric_limits<float>::lowest();
  }
  if (D == 2)
    g.minC[2] = g.maxC[2] = 0.f;
  float tmin[kMaxHostThreads][3], tmax[kMaxHostThreads][3];
  for (unsigned t = 0; t < kMaxHostThreads; ++t)
    for (int k = 0; k < 3; ++k) {
      tmin[t][k] = std::numeric_limits<float>::max();
      tmax[t][k] = std::numeric_limits<float>::lowest();
    }
  parallel_ranges(n, [&](unsigned t, uint32_t b, uint32_t e) {
    float lmin[3] = {tmin[t][0], tmin[t][1], tmin[t][2]}, lmax[3] = {tmax[t][0], tmax[t][1], tmax[t][2]};
    for (uint32_t i = b; i < e; ++i) {
      const float *p = pts + 3 * (size_t)i;
      float *d = &g.disk4[4 * (size_t)i];
      d[0] = p[0];
      d[1] = p[1];
      d[2] = D == 2 ? 0.f : p[2];
      d[3] = g.diskRadius;
      g.points3[3 * (size_t)i] = p[0];
      g.points3[3 * (size_t)i + 1] = p[1];
      g.points3[3 * (size_t)i + 2] = D == 2 ? 0.f : p[2]; // (2-D: the z column is ignored, rayGeometryDisk.hpp:148-151)
      for (int k = 0; k < D; ++k) { // (thread-local: the shared arrays would ping-pong between cores)
        lmin[k] = std::min(lmin[k], p[k]);
        lmax[k] = std::max(lmax[k], p[k]);
      }
      g.normal3[3 * (size_t)i + 0] = nrm[3 * (size_t)i + 0];
      g.normal3[3 * (size_t)i + 1] = nrm[3 * (size_t)i + 1];
      g.normal3[3 * (size_t)i + 2] = D == 2 ? 0.f : nrm[3 * (size_t)i + 2];
    }
    for (int k = 0; k < 3; ++k) {
      tmin[t][k] = lmin[k];
      tmax[t][k] = lmax[k];
    }
  });
  for (unsigned t = 0; t < kMaxHostThreads; ++t)
    for (int k = 0; k < D; ++k) { // (min/max: any order gives the same result)
      g.minC[k] = std::min(g.minC[k], tmin[t][k]);
      g.maxC[k] = std::max(g.maxC[k], tmax[t][k]);
    }
  if (g.materialIds.size() != n)
    g.materialIds.assign(n, 0);
  // the neighbourhood (rayGeometryDisk.hpp:191-192, radius = 2 * disk radius) is built
  // on the device with the BVH (vr_setup.hip); the host version is a validation path
  g.nbOff.clear();
  g.nbIds.clear();
  g.verts.clear();
  g.tris.clear();
  g.triAreas.clear();
}

void host_set_triangles(HostGeometry &g, const float *verts, uint32_t nv, const uint32_t *tris, uint32_t nt,
                        float gridDelta, int D) {
  g.D = D;
  g.geo = 1;
  g.numPrims = nt;
  g.gridDelta = gridDelta;
  g.diskRadius = 0.f;
  g.verts.assign(verts, verts + (size_t)nv * 3);
  g.tris.assign(tris, tris + (size_t)nt * 3);
  // rayMesh.hpp:12-25: bounding box over all nodes
  for (int k = 0; k < 3; ++k) {
    g.minC[k] = nv ? verts[k] : 0.f;
    g.maxC[k] = nv ? verts[k] : 0.f;
  }
  for (uint32_t i = 0; i < nv; ++i)
    for (int k = 0; k < 3; ++k) {
      g.minC[k] = std::min(g.minC[k], verts[3 * (size_t)i + k]);
      g.maxC[k] = std::max(g.maxC[k], verts[3 * (size_t)i + k]);
    }
  g.normal3.resize((size_t)nt * 3);
  g.triAreas.resize(nt);
  for (uint32_t i = 0; i < nt; ++i) {
    const float *a = &verts[3 * (size_t)tris[3 * (size_t)i]];
    const float *b = &verts[3 * (size_t)tris[3 * (size_t)i + 1]];
    const float *c = &verts[3 * (size_t)tris[3 * (size_t)i + 2]];
    const F3 v0{a[0], a[1], a[2]}, v1{b[0], b[1], b[2]}, v2{c[0], c[1], c[2]};
    F3 nn = cross3(sub3(v1, v0), sub3(v2, v0));
    // rayGeometryTriangle.hpp:62-75
    if (D == 2)
      g.triAreas[i] = (float)(0.5 * norm3((i % 2 == 0) ? sub3(v1, v0) : sub3(v2, v0)));
    else
      g.triAreas[i] = (float)(0.5 * norm3(nn));
    normalize3(nn); // rayMesh.hpp:108-110
    g.normal3[3 * (size_t)i + 0] = nn.x;
    g.normal3[3 * (size_t)i + 1] = nn.y;
    g.normal3[3 * (size_t)i + 2] = nn.z;
  }
  if (g.materialIds.size() != nt)
    g.materialIds.assign(nt, 0);
  g.disk4.clear();
  g.points3.clear();
  g.nbOff.assign((size_t)nt + 1, 0u);
  g.nbIds.clear();
}

// ---------------------------------------------------------------------------
// neighbourhood: all pairs with per-axis |d| <= dist (first D axes) and
// |p1-p2|^2 <= dist^2 (rayPointNeighborhood.hpp:287-298).  Sort-based uniform
// grid with cell size = dist.
// ---------------------------------------------------------------------------
void host_neighbors(int D, const float *pts3, uint32_t n, float dist, const float *minC, std::vector<uint32_t> &off,
                    std::vector<uint32_t> &ids) {
  off.assign((size_t)n + 1, 0u);
  ids.clear();
  if (n == 0 || !(dist > 0.f))
    return;
  const float dist2 = dist * dist;
  const float invCell = 1.f / dist;
  auto cellOf = [&](uint32_t i, int k) -> int64_t {
    if (k >= D)
      return 0;
    return (int64_t)std::floor((pts3[3 * (size_t)i + k] - minC[k]) * invCell);
  };
  auto keyOf = [](int64_t cx, int64_t cy, int64_t cz) -> uint64_t {
    return ((uint64_t)(cx + 1) << 42) | ((uint64_t)(cy + 1) << 21) | (uint64_t)(cz + 1);
  };
  std::vector<uint64_t> key(n);
  for (uint32_t i = 0; i < n; ++i)
    key[i] = keyOf(cellOf(i, 0), cellOf(i, 1), cellOf(i, 2));
  std::vector<uint32_t> perm(n);
  std::iota(perm.begin(), perm.end(), 0u);
  std::sort(perm.begin(), perm.end(), [&](uint32_t a, uint32_t b) { return key[a] < key[b] || (key[a] == key[b] && a < b); });
  std::vector<uint64_t> skey(n);
  for (uint32_t i = 0; i < n; ++i)
    skey[i] = key[perm[i]];
  auto near = [&](uint32_t i, uint32_t j) {
    const float *a = pts3 + 3 * (size_t)i, *b = pts3 + 3 * (size_t)j;
    for (int k = 0; k < D; ++k)
      if (std::abs(a[k] - b[k]) > dist)
        return false;
    const float dx = a[0] - b[0], dy = a[1] - b[1], dz = a[2] - b[2];
    return ((dx * dx + dy * dy) + dz * dz) <= dist2;
  };
  const int zlo = D == 3 ? -1 : 0, zhi = D == 3 ? 1 : 0;
  std::vector<uint32_t> tmp;
  // two passes: count, then fill
  for (int pass = 0; pass < 2; ++pass) {
    if (pass == 1) {
      uint32_t acc = 0;
      for (uint32_t i = 0; i < n; ++i) {
        uint32_t c = off[i + 1];
        off[i] = acc;
        acc += c;
      }
      off[n] = acc;
      ids.resize(acc);
    }
    for (uint32_t i = 0; i < n; ++i) {
      const int64_t cx = cellOf(i, 0), cy = cellOf(i, 1), cz = cellOf(i, 2);
      tmp.clear();
      for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
          for (int dz = zlo; dz <= zhi; ++dz) {
            if (cx + dx < -1 || cy + dy < -1 || cz + dz < -1)
              continue;
            const uint64_t k = keyOf(cx + dx, cy + dy, cz + dz);
            auto it = std::lower_bound(skey.begin(), skey.end(), k);
            for (size_t s = it - skey.begin(); s < n && skey[s] == k; ++s) {
              const uint32_t j = perm[s];
              if (j != i && near(i, j))
                tmp.push_back(j);
            }
          }
      if (pass == 0) {
        off[i + 1] = (uint32_t)tmp.size();
      } else {
        std::sort(tmp.begin(), tmp.end());
        std::copy(tmp.begin(), tmp.end(), ids.begin() + off[i]);
      }
    }
  }
}

// ---------------------------------------------------------------------------
// rayUtil.hpp:104-143
void host_adjust_bbox(float *lo, float *hi, int D, int direction, float pad) {
  if (D == 2) {
    lo[2] -= pad;
    hi[2] += pad;
  }
  switch (direction) {
  case 0: hi[0] += 2 * pad; break;
  case 1: lo[0] -= 2 * pad; break;
  case 2: hi[1] += 2 * pad; break;
  case 3: lo[1] -= 2 * pad; break;
  case 4: hi[2] += 2 * pad; break;
  case 5: lo[2] -= 2 * pad; break;
  }
}

// rayUtil.hpp:145-202: {rayDir, firstDir, secondDir, minMax, posNeg}
std::array<int, 5> host_trace_settings(int direction) {
  switch (direction) {
  case 0: return {0, 1, 2, 1, -1};
  case 1: return {0, 1, 2, 0, 1};
  case 2: return {1, 0, 2, 1, -1};
  case 3: return {1, 0, 2, 0, 1};
  case 4: return {2, 0, 1, 1, -1};
  default: return {2, 0, 1, 0, 1};
  }
}

// rayBoundary.hpp:174-236
void host_build_walls(const float *lo, const float *hi, int firstDir, int secondDir, Tri *wall) {
  const float V[8][3] = {{lo[0], lo[1], lo[2]}, {hi[0], lo[1], lo[2]}, {hi[0], hi[1], lo[2]}, {lo[0], hi[1], lo[2]},
                         {lo[0], lo[1], hi[2]}, {hi[0], lo[1], hi[2]}, {hi[0], hi[1], hi[2]}, {lo[0], hi[1], hi[2]}};
  static const unsigned P[3][4][3] = {{{0, 3, 7}, {0, 7, 4}, {6, 2, 1}, {6, 1, 5}},
                                      {{0, 4, 5}, {0, 5, 1}, {6, 7, 3}, {6, 3, 2}},
                                      {{0, 1, 2}, {0, 2, 3}, {6, 5, 4}, {6, 4, 7}}};
  auto mk = [&](const unsigned *t, Tri &w) {
    const float *a = V[t[0]], *b = V[t[1]], *c = V[t[2]];
    for (int k = 0; k < 3; ++k) {
      w.v0[k] = a[k];
      w.e1[k] = a[k] - b[k];
      w.e2[k] = c[k] - a[k];
    }
    // Ng = cross(e2, e1)
    w.Ng[0] = w.e2[1] * w.e1[2] - w.e2[2] * w.e1[1];
    w.Ng[1] = w.e2[2] * w.e1[0] - w.e2[0] * w.e1[2];
    w.Ng[2] = w.e2[0] * w.e1[1] - w.e2[1] * w.e1[0];
  };
  for (int i = 0; i < 4; ++i) {
    mk(P[firstDir][i], wall[i]);
    mk(P[secondDir][i], wall[i + 4]);
  }
}

// rayUtil.hpp:287-321
void host_orthonormal_basis(const float *v, float *basis9) {
  F3 u{v[0], v[1], v[2]};
  const float len2 = dot3(u, u);
  const float invLen = 1.f / std::sqrt(len2);
  u = F3{u.x * invLen, u.y * invLen, u.z * invLen};
  F3 h = std::abs(u.x) > std::abs(u.z) ? F3{-u.y, u.x, 0.f} : F3{0.f, -u.z, u.y};
  normalize3(h);
  const F3 w = cross3(u, h);
  const float b[9] = {u.x, u.y, u.z, h.x, h.y, h.z, w.x, w.y, w.z};
  std::memcpy(basis9, b, sizeof(b));
}

// ---------------------------------------------------------------------------
// Disk areas clipped by the x/y walls (rayGeometryDisk.hpp:266-354 and
// rayDiskBoundingBoxIntersector.hpp:39-432): the HOST evaluation of vr_area.hpp, the source
// the device kernel is compiled from (validation path, VR_HOST_BUILD=1).
// ---------------------------------------------------------------------------
void host_disk_areas(const HostGeometry &g, const AreaParams &p, std::vector<float> &areas) {
  const uint32_t n = g.numPrims;
  areas.assign(n, 0.f);
  if (g.geo != 0)
    return;
  parallel_ranges(n, [&](unsigned, uint32_t rb, uint32_t re) {
    for (uint32_t i = rb; i < re; ++i)
      areas[i] = disk_exposed_area(p, &g.disk4[4 * (size_t)i], &g.normal3[3 * (size_t)i]);
  });
}

float host_sort_plane(const HostGeometry &g, int axis, float fallback, float *modeShare) {
  const float lo = g.minC[axis], hi = g.maxC[axis];
  if (modeShare)
    *modeShare = 1.f;
  if (g.numPrims == 0 || !(hi > lo))
    return g.numPrims ? lo : fallback;
  constexpr int SL = 256;
  std::vector<double> w(SL, 0.0), wh(SL, 0.0);
  const double inv = SL / ((double)hi - (double)lo);
  // per-thread histograms, merged in thread order (the plane only orders work: its last bits
  // have no influence on any result)
  std::vector<double> tw((size_t)kMaxHostThreads * SL, 0.0), twh((size_t)kMaxHostThreads * SL, 0.0);
  parallel_ranges(g.numPrims, [&](unsigned t, uint32_t rb, uint32_t re) {
  double *w = &tw[(size_t)t * SL], *wh = &twh[(size_t)t * SL];
  for (uint32_t i = rb; i < re; ++i) {
    double h, a;
    if (g.geo == 0) {
      const float *d = &g.disk4[4 * (size_t)i];
      const float *n = &g.normal3[3 * (size_t)i];
      const double nn = std::sqrt((double)n[0] * n[0] + (double)n[1] * n[1] + (double)n[2] * n[2]);
      h = d[axis];
      a = nn > 0. ? (double)d[3] * d[3] * std::fabs((double)n[axis]) / nn : 0.;
    } else {
      const float *v0 = &g.verts[3 * (size_t)g.tris[3 * (size_t)i]];
      const float *v1 = &g.verts[3 * (size_t)g.tris[3 * (size_t)i + 1]];
      const float *v2 = &g.verts[3 * (size_t)g.tris[3 * (size_t)i + 2]];
      const int a1 = (axis + 1) % 3, a2 = (axis + 2) % 3;
      const double e1[2] = {(double)v1[a1] - v0[a1], (double)v1[a2] - v0[a2]};
      const double e2[2] = {(double)v2[a1] - v0[a1], (double)v2[a2] - v0[a2]};
      h = ((double)v0[axis] + v1[axis] + v2[axis]) / 3.;
      a = 0.5 * std::fabs(e1[0] * e2[1] - e1[1] * e2[0]); // area of the projection along the axis
    }
    int k = (int)((h - lo) * inv);
    k = k < 0 ? 0 : (k >= SL ? SL - 1 : k);
    w[k] += a;
    wh[k] += a * h;
  }
  });
  for (unsigned t = 0; t < kMaxHostThreads; ++t)
    for (int k = 0; k < SL; ++k) {
      w[k] += tw[(size_t)t * SL + k];
      wh[k] += twh[(size_t)t * SL + k];
    }
  int best = 0;
  double total = w[0];
  for (int k = 1; k < SL; ++k) {
    total += w[k];
    if (w[k] > w[best])
      best = k;
  }
  if (modeShare && total > 0.)
    *modeShare = (float)(w[best] / total);
  return w[best] > 0. ? (float)(wh[best] / w[best]) : fallback;
}

// ---------------------------------------------------------------------------
// LBVH (host): 63-bit Morton codes of box centres, sort, split at the highest
// differing bit, leaves of <= VR_LEAF_MAX primitives, nodes emitted in
// pre-order so that left child = parent + 1; escape links in a second pass.
// ---------------------------------------------------------------------------
namespace {
inline uint64_t spread21(uint64_t v) {
  v &= 0x1FFFFFull;
  v = (v | v << 32) & 0x1F00000000FFFFull;
  v = (v | v << 16) & 0x1F0000FF0000FFull;
  v = (v | v << 8) & 0x100F00F00F00F00Full;
  v = (v | v << 4) & 0x10C30C30C30C30C3ull;
  v = (v | v << 2) & 0x1249249249249249ull;
  return v;
}

struct Builder {
  const std::vector<float> &box; // 6 per prim (sorted order)
  const std::vector<uint64_t> &code;
  std::vector<float> &nodes;
  std::vector<uint32_t> right; // right child per node (internal), 0 for leaves
  uint32_t leaves = 0, maxDepth = 0;

  uint32_t build(uint32_t first, uint32_t last, uint32_t depth) { // inclusive range
    const uint32_t me = (uint32_t)(nodes.size() / 8);
    nodes.resize(nodes.size() + 8);
    right.push_back(0);
    maxDepth = std::max(maxDepth, depth);
    const uint32_t cnt = last - first + 1;
    float lo[3], hi[3];
    if (cnt <= (uint32_t)VR_LEAF_MAX) {
      for (int k = 0; k < 3; ++k) {
        lo[k] = FLT_MAX;
        hi[k] = -FLT_MAX;
      }
      for (uint32_t i = first; i <= last; ++i)
        for (int k = 0; k < 3; ++k) {
          lo[k] = std::min(lo[k], box[6 * (size_t)i + k]);
          hi[k] = std::max(hi[k], box[6 * (size_t)i + 3 + k]);
        }
      const uint32_t link = VR_LEAF | (cnt << 27) | first;
      float *nd = &nodes[8 * (size_t)me];
      std::memcpy(nd, lo, 12);
      std::memcpy(nd + 3, &link, 4);
      std::memcpy(nd + 4, hi, 12);
      ++leaves;
      return me;
    }
    // split position: last index of the left part
    uint32_t split;
    const uint64_t cf = code[first], cl = code[last];
    if (cf == cl) {
      split = first + (cnt - 1) / 2;
    } else {
      const int prefix = __builtin_clzll(cf ^ cl);
      split = first;
      uint32_t step = last - first;
      do {
        step = (step + 1) >> 1;
        const uint32_t ns = split + step;
        if (ns < last) {
          const uint64_t c = code[ns];
          if (c == cf || __builtin_clzll(cf ^ c) > prefix)
            split = ns;
        }
      } while (step > 1);
    }
    const uint32_t l = build(first, split, depth + 1);
    const uint32_t r = build(split + 1, last, depth + 1);
    right[me] = r;
    float *nd = &nodes[8 * (size_t)me];
    const float *a = &nodes[8 * (size_t)l], *b = &nodes[8 * (size_t)r];
    for (int k = 0; k < 3; ++k) {
      nd[k] = std::min(a[k], b[k]);
      nd[4 + k] = std::max(a[4 + k], b[4 + k]);
    }
    std::memcpy(nd + 3, &l, 4);
    return me;
  }
};
} // namespace

namespace {
// Binned-SAH top-down builder over the same (padded) primitive boxes: the permutation is refined in place, a leaf is a
// contiguous run of it, nodes are emitted in pre-order like the LBVH's — the rest of the pipeline cannot tell the two
// trees apart.  (VR_HOST_BUILD=2; tree quality experiment for the per-lane walks of the bounce-heavy workloads)
struct SahBuilder {
  const std::vector<float> &box; // 6 per primitive, ORIGINAL order
  std::vector<uint32_t> &order;  // permutation being refined
  std::vector<float> &nodes;
  std::vector<uint32_t> right;
  uint32_t leafMax = VR_LEAF_MAX, leaves = 0, maxDepth = 0;

  static float area(const float *lo, const float *hi) {
    const float dx = std::max(hi[0] - lo[0], 0.f), dy = std::max(hi[1] - lo[1], 0.f), dz = std::max(hi[2] - lo[2], 0.f);
    return dx * dy + dy * dz + dz * dx;
  }
  uint32_t build(uint32_t first, uint32_t last, uint32_t depth) { // inclusive range of `order`
    const uint32_t me = (uint32_t)(nodes.size() / 8);
    nodes.resize(nodes.size() + 8);
    right.push_back(0);
    maxDepth = std::max(maxDepth, depth);
    const uint32_t cnt = last - first + 1;
    float lo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, hi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    float clo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, chi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
    for (uint32_t i = first; i <= last; ++i) {
      const float *b = &box[6 * (size_t)order[i]];
      for (int k = 0; k < 3; ++k) {
        lo[k] = std::min(lo[k], b[k]);
        hi[k] = std::max(hi[k], b[3 + k]);
        const float c = 0.5f * (b[k] + b[3 + k]);
        clo[k] = std::min(clo[k], c);
        chi[k] = std::max(chi[k], c);
      }
    }
    if (cnt <= leafMax) {
      const uint32_t link = VR_LEAF | (cnt << 27) | first;
      float *nd = &nodes[8 * (size_t)me];
      std::memcpy(nd, lo, 12);
      std::memcpy(nd + 3, &link, 4);
      std::memcpy(nd + 4, hi, 12);
      ++leaves;
      return me;
    }
    // best binned split over the three axes
    constexpr int NB = 16;
    int bestAxis = -1, bestBin = 0;
    float bestCost = FLT_MAX;
    for (int ax = 0; ax < 3; ++ax) {
      const float ext = chi[ax] - clo[ax];
      if (!(ext > 0.f))
        continue;
      const float scale = NB / ext;
      uint32_t bc[NB] = {0};
      float blo[NB][3], bhi[NB][3];
      for (int b = 0; b < NB; ++b)
        for (int k = 0; k < 3; ++k)
          blo[b][k] = FLT_MAX, bhi[b][k] = -FLT_MAX;
      for (uint32_t i = first; i <= last; ++i) {
        const float *b = &box[6 * (size_t)order[i]];
        int bi = (int)((0.5f * (b[ax] + b[3 + ax]) - clo[ax]) * scale);
        bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
        ++bc[bi];
        for (int k = 0; k < 3; ++k) {
          blo[bi][k] = std::min(blo[bi][k], b[k]);
          bhi[bi][k] = std::max(bhi[bi][k], b[3 + k]);
        }
      }
      float rA[NB];
      uint32_t rN[NB];
      {
        float l[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, h[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
        uint32_t c = 0;
        for (int b = NB - 1; b > 0; --b) {
          for (int k = 0; k < 3; ++k)
            l[k] = std::min(l[k], blo[b][k]), h[k] = std::max(h[k], bhi[b][k]);
          c += bc[b];
          rA[b] = c ? area(l, h) : 0.f;
          rN[b] = c;
        }
      }
      float l[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, h[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
      uint32_t c = 0;
      for (int b = 0; b < NB - 1; ++b) { // split between bin b and b + 1
        for (int k = 0; k < 3; ++k)
          l[k] = std::min(l[k], blo[b][k]), h[k] = std::max(h[k], bhi[b][k]);
        c += bc[b];
        if (c == 0 || rN[b + 1] == 0)
          continue;
        const float cost = area(l, h) * (float)c + rA[b + 1] * (float)rN[b + 1];
        if (cost < bestCost)
          bestCost = cost, bestAxis = ax, bestBin = b;
      }
    }
    uint32_t mid; // first index of the right part
    if (bestAxis < 0) {
      mid = first + cnt / 2; // (all centres coincide)
    } else {
      const float scale = NB / (chi[bestAxis] - clo[bestAxis]);
      auto it = std::partition(order.begin() + first, order.begin() + last + 1, [&](uint32_t o) {
        const float *b = &box[6 * (size_t)o];
        int bi = (int)((0.5f * (b[bestAxis] + b[3 + bestAxis]) - clo[bestAxis]) * scale);
        bi = bi < 0 ? 0 : (bi >= NB ? NB - 1 : bi);
        return bi <= bestBin;
      });
      mid = (uint32_t)(it - order.begin());
      if (mid == first || mid > last)
        mid = first + cnt / 2;
    }
    const uint32_t l = build(first, mid - 1, depth + 1);
    const uint32_t r = build(mid, last, depth + 1);
    right[me] = r;
    float *nd = &nodes[8 * (size_t)me];
    std::memcpy(nd, lo, 12);
    std::memcpy(nd + 3, &l, 4);
    std::memcpy(nd + 4, hi, 12);
    return me;
  }
};
} // namespace

void host_build_bvh(const HostGeometry &g, Bvh &bvh) {
  const uint32_t n = g.numPrims;
  bvh.nodes.clear();
  bvh.order.clear();
  bvh.numNodes = bvh.numLeaves = bvh.maxDepth = 0;
  if (n == 0)
    return;
  // primitive boxes
  std::vector<float> box((size_t)n * 6);
  float slo[3] = {FLT_MAX, FLT_MAX, FLT_MAX}, shi[3] = {-FLT_MAX, -FLT_MAX, -FLT_MAX};
  for (uint32_t i = 0; i < n; ++i) {
    float *b = &box[6 * (size_t)i];
    if (g.geo == 0) {
      const float *d = &g.disk4[4 * (size_t)i];
      F3 nn{g.normal3[3 * (size_t)i], g.normal3[3 * (size_t)i + 1], g.normal3[3 * (size_t)i + 2]};
      normalize3(nn);
      const float nv[3] = {nn.x, nn.y, nn.z};
      for (int k = 0; k < 3; ++k) {
        // extent of an oriented disc along axis k: r * sqrt(1 - n_k^2)
        const float h = d[3] * std::sqrt(std::max(0.f, 1.f - nv[k] * nv[k])) * 1.0001f;
        b[k] = d[k] - h;
        b[3 + k] = d[k] + h;
      }
    } else {
      const float *v0 = &g.verts[3 * (size_t)g.tris[3 * (size_t)i]];
      const float *v1 = &g.verts[3 * (size_t)g.tris[3 * (size_t)i + 1]];
      const float *v2 = &g.verts[3 * (size_t)g.tris[3 * (size_t)i + 2]];
      for (int k = 0; k < 3; ++k) {
        b[k] = std::min(v0[k], std::min(v1[k], v2[k]));
        b[3 + k] = std::max(v0[k], std::max(v1[k], v2[k]));
      }
    }
    for (int k = 0; k < 3; ++k) {
      slo[k] = std::min(slo[k], b[k]);
      shi[k] = std::max(shi[k], b[3 + k]);
    }
  }
  // pad every box by more than the rounding error of the device slab test
  float scale = 0.f;
  for (int k = 0; k < 3; ++k)
    scale = std::max(scale, std::max(std::fabs(slo[k]), std::fabs(shi[k])));
  const float pad = 4e-6f * std::max(scale, 1e-3f);
  for (size_t i = 0; i < box.size(); i += 6)
    for (int k = 0; k < 3; ++k) {
      box[i + k] -= pad;
      box[i + 3 + k] += pad;
    }
  if (const char *hb = std::getenv("VR_HOST_BUILD"); hb && std::atoi(hb) == 2) { // binned-SAH tree (experiment)
    bvh.order.resize(n);
    std::iota(bvh.order.begin(), bvh.order.end(), 0u);
    bvh.nodes.reserve((size_t)n * 8);
    SahBuilder B{box, bvh.order, bvh.nodes, {}, (uint32_t)(g.geo == 0 ? VR_LEAF_MAX : 3), 0, 0};
    B.right.reserve(n);
    B.build(0, n - 1, 0);
    bvh.numNodes = (uint32_t)(bvh.nodes.size() / 8);
    bvh.numLeaves = B.leaves;
    bvh.maxDepth = B.maxDepth;
    std::vector<uint32_t> esc(bvh.numNodes, VR_END);
    for (uint32_t i = 0; i < bvh.numNodes; ++i) {
      uint32_t link;
      std::memcpy(&link, &bvh.nodes[8 * (size_t)i + 3], 4);
      if (!(link & VR_LEAF)) {
        esc[link] = B.right[i];
        esc[B.right[i]] = esc[i];
      }
      std::memcpy(&bvh.nodes[8 * (size_t)i + 7], &esc[i], 4);
    }
    return;
  }
  // Morton codes of the box centres
  std::vector<uint64_t> code(n);
  float inv[3]; // (cells in the scene box's proportions, at most VR_MORTON_ANISO : 1 — as morton_kernel, vr_setup.hip)
  {
    float aniso = VR_MORTON_ANISO;
    if (const char *e = std::getenv("VR_MORTON_ANISO"))
      aniso = std::max(1.f, (float)std::atof(e));
    const float extMax = std::max(std::max(shi[0] - slo[0], shi[1] - slo[1]), shi[2] - slo[2]);
    for (int k = 0; k < 3; ++k) {
      const float ext = std::max(shi[k] - slo[k], extMax / aniso);
      inv[k] = ext > 0.f ? 2097151.0f / ext : 0.f;
    }
  }
  for (uint32_t i = 0; i < n; ++i) {
    const float *b = &box[6 * (size_t)i];
    uint64_t q[3];
    for (int k = 0; k < 3; ++k) {
      float c = (0.5f * (b[k] + b[3 + k]) - slo[k]) * inv[k];
      c = std::min(std::max(c, 0.f), 2097151.0f);
      q[k] = (uint64_t)c;
    }
    code[i] = (spread21(q[0]) << 2) | (spread21(q[1]) << 1) | spread21(q[2]);
  }
  bvh.order.resize(n);
  std::iota(bvh.order.begin(), bvh.order.end(), 0u);
  std::sort(bvh.order.begin(), bvh.order.end(),
            [&](uint32_t a, uint32_t b) { return code[a] < code[b] || (code[a] == code[b] && a < b); });
  std::vector<uint64_t> scode(n);
  std::vector<float> sbox((size_t)n * 6);
  for (uint32_t q = 0; q < n; ++q) {
    scode[q] = code[bvh.order[q]];
    std::memcpy(&sbox[6 * (size_t)q], &box[6 * (size_t)bvh.order[q]], 24);
  }
  bvh.nodes.reserve((size_t)n * 8);
  Builder B{sbox, scode, bvh.nodes, {}, 0, 0};
  B.right.reserve(n);
  B.build(0, n - 1, 0);
  bvh.numNodes = (uint32_t)(bvh.nodes.size() / 8);
  bvh.numLeaves = B.leaves;
  bvh.maxDepth = B.maxDepth;
  // escape links: parents precede children in pre-order
  std::vector<uint32_t> esc(bvh.numNodes, VR_END);
  for (uint32_t i = 0; i < bvh.numNodes; ++i) {
    uint32_t link;
    std::memcpy(&link, &bvh.nodes[8 * (size_t)i + 3], 4);
    if (!(link & VR_LEAF)) {
      esc[link] = B.right[i];     // left child -> sibling
      esc[B.right[i]] = esc[i];   // right child -> parent's escape
    }
    std::memcpy(&bvh.nodes[8 * (size_t)i + 7], &esc[i], 4);
  }
}

void host_pack_prims(const HostGeometry &g, const Bvh &bvh, std::vector<float> &prims) {
  const uint32_t n = g.numPrims;
  const size_t rec = g.geo == 0 ? 8 : 16;
  prims.assign((size_t)n * rec, 0.f);
  for (uint32_t q = 0; q < n; ++q) {
    const uint32_t o = bvh.order[q];
    float *r = &prims[rec * (size_t)q];
    if (g.geo == 0) {
      std::memcpy(r, &g.disk4[4 * (size_t)o], 16);
      std::memcpy(r + 4, &g.normal3[3 * (size_t)o], 12);
      std::memcpy(r + 7, &o, 4);
    } else {
      const float *a = &g.verts[3 * (size_t)g.tris[3 * (size_t)o]];
      const float *b = &g.verts[3 * (size_t)g.tris[3 * (size_t)o + 1]];
      const float *c = &g.verts[3 * (size_t)g.tris[3 * (size_t)o + 2]];
      float e1[3], e2[3], Ng[3];
      for (int k = 0; k < 3; ++k) {
        e1[k] = a[k] - b[k];
        e2[k] = c[k] - a[k];
      }
      Ng[0] = e2[1] * e1[2] - e2[2] * e1[1];
      Ng[1] = e2[2] * e1[0] - e2[0] * e1[2];
      Ng[2] = e2[0] * e1[1] - e2[1] * e1[0];
      std::memcpy(r, a, 12);
      std::memcpy(r + 3, &o, 4);
      std::memcpy(r + 4, e1, 12);
      std::memcpy(r + 8, e2, 12);
      std::memcpy(r + 12, Ng, 12);
      r[7] = g.normal3[3 * (size_t)o];
      r[11] = g.normal3[3 * (size_t)o + 1];
      r[15] = g.normal3[3 * (size_t)o + 2];
    }
  }
}

} // namespace vr
