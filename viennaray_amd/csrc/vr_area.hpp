// vr_area.hpp — exposed area of a disk inside the lateral bounding box (flux normalisation).
//
// Restates rayGeometryDisk.hpp:266-354 (computeDiskAreas) and
// rayDiskBoundingBoxIntersector.hpp:39-432 (DiskBoundingBoxXYIntersector).  ONE source for the
// device kernel (vr_setup.hip: disk_areas_kernel, one thread per disk) and for the host
// validation path (VR_HOST_BUILD=1): float operations in the reference's order, no contraction,
// glibc's acosf / sinf reproduced bit for bit (vr_libm.hpp), IEEE sqrt and division — so both
// give the bits the reference's CPU loop gives, and the CPU oracle (an independent restatement
// on the real glibc) checks them.
//
// The four walls are visited clockwise (right, bottom, left, top) like the reference; each wall
// is described directly in world coordinates instead of through the reference's swap/reflect
// transforms: axis, outward sign, plane coordinate, inward normal, and the corner it shares
// with the NEXT wall.
#pragma once
#include "vr_libm.hpp"

namespace vr {

struct A3 {
  float x, y, z;
};
VR_HD float a_dot(const A3 &a, const A3 &b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
VR_HD A3 a_cross(const A3 &a, const A3 &b) {
  return A3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
VR_HD A3 a_sub(const A3 &a, const A3 &b) { return A3{a.x - b.x, a.y - b.y, a.z - b.z}; }
VR_HD float a_norm(const A3 &a) { return __builtin_sqrtf(a_dot(a, a)); }
VR_HD void a_normalize(A3 &a) {
  const float n = a_norm(a);
  if (n <= 0.f)
    return;
  a.x /= n;
  a.y /= n;
  a.z /= n;
}

constexpr double VR_PI = 3.14159265358979323846;
constexpr float VR_FLT_MAX = 3.402823466e+38f;

// what the area computation needs besides the disk itself
struct AreaParams {
  int D;
  int firstDir, secondDir;   // lateral axes of the trace settings
  int bcFirst, bcSecond;     // boundary condition of the wall pair on firstDir / secondDir (by AXIS,
                             // rayGeometryDisk.hpp:281-284; axis 2 uses entry 1)
  float minC[3], maxC[3];    // bounding box of the disk CENTRES (geometry bbox, not the adjusted one)
};

VR_HD float disk_area_inside_xy(const float *disk, const float *nrm, float lx, float ly, float hx, float hy) {
  const float xx = disk[0], yy = disk[1], radius = disk[3];
  A3 dn{nrm[0], nrm[1], nrm[2]};
  a_normalize(dn);
  const float full = (float)(radius * radius * VR_PI);
  if ((lx <= xx - radius && xx + radius <= hx) && (ly <= yy - radius && yy + radius <= hy))
    return full;
  if ((xx + radius <= lx || hx <= xx - radius) || (yy + radius <= ly || hy <= yy - radius))
    return 0.f;
  // walls: {axis, outward sign, plane coordinate, inward normal, corner shared with the next wall}
  const int wAxis[4] = {0, 1, 0, 1};
  const float wSgn[4] = {1.f, -1.f, -1.f, 1.f};
  const float wW[4] = {hx, ly, lx, hy};
  const A3 wIn[4] = {A3{-1, 0, 0}, A3{0, 1, 0}, A3{1, 0, 0}, A3{0, -1, 0}};
  const float wCx[4] = {hx, hx, lx, lx}, wCy[4] = {hy, ly, ly, hy};
  float approach[4] = {0.f, 0.f, 0.f, 0.f};
  for (int k = 0; k < 4; ++k) {
    const float c = wSgn[k] * disk[wAxis[k]]; // coordinate in the wall's outward frame
    const float H = wSgn[k] * wW[k];
    const float nb = wAxis[k] == 0 ? dn.y : dn.x;
    const float xterm = radius * __builtin_sqrtf(dn.z * dn.z + nb * nb);
    float a;
    if (c + xterm <= H)
      a = VR_FLT_MAX;
    else if (c - xterm >= H)
      a = -VR_FLT_MAX;
    else if (xterm <= 1e-9)
      a = VR_FLT_MAX;
    else
      a = (H - c) * radius / xterm;
    approach[k] = a;
    if (a < -radius)
      return 0.f; // fully outside (later entries stay 0 in the reference, then it returns 0)
  }
  float area = 0.f;
  for (int k = 0; k < 4; ++k) {
    const float d = approach[k];
    if (-radius < d && d < radius) {
      const float angle = 2 * glibc_acosf(d / radius);
      area += radius * radius / 2 * (angle - glibc_sinf(angle));
    }
  }
  const A3 c{disk[0], disk[1], disk[2]};
  for (int k = 0; k < 4; ++k) {
    const int k2 = (k + 1) % 4;
    const float d1 = approach[k], d2 = approach[k2];
    if (!(-radius < d1 && d1 < radius && -radius < d2 && d2 < radius))
      continue;
    const A3 n1 = wIn[k], n2 = wIn[k2];
    A3 i1 = a_cross(dn, n1);
    a_normalize(i1);
    A3 i2 = a_cross(dn, n2);
    a_normalize(i2);
    if (a_dot(i1, n2) >= 0)
      i1 = A3{-i1.x, -i1.y, -i1.z};
    if (a_dot(i2, n1) >= 0)
      i2 = A3{-i2.x, -i2.y, -i2.z};
    const float px = wCx[k2], py = wCy[k2];
    const A3 ip{px, py, (dn.x * c.x + dn.y * c.y + dn.z * c.z - dn.x * px - dn.y * py) / dn.z};
    if (a_norm(a_sub(c, ip)) >= radius)
      continue;
    // The reference derives each wall's normal from a triangle spanning the wall
    // (rayDiskBoundingBoxIntersector.hpp:124-135); on a bounding box without extent along the
    // wall that triangle is degenerate, its normalised normal is 0/0 and the area comes out
    // NaN (single disk, one row of disks).  Reproduced rather than "fixed".
    const bool flat1 = wAxis[k] == 0 ? hy == ly : hx == lx;
    const bool flat2 = wAxis[k2] == 0 ? hy == ly : hx == lx;
    if (flat1 || flat2)
      return vr_asfloat(0x7fc00000u);
    A3 q[2];
    for (int j = 0; j < 2; ++j) {
      const A3 &iDir = j == 0 ? i1 : i2;
      const float d = j == 0 ? d1 : d2;
      const float ca = a_dot(a_sub(c, ip), iDir);
      const A3 cp{ip.x + ca * iDir.x, ip.y + ca * iDir.y, ip.z + ca * iDir.z};
      const float thc = __builtin_sqrtf(radius * radius - d * d);
      q[j] = A3{cp.x + iDir.x * thc, cp.y + iDir.y * thc, cp.z + iDir.z * thc};
    }
    const A3 c1 = a_sub(q[0], c), c2 = a_sub(q[1], c);
    const float angle = glibc_acosf(a_dot(c1, c2) / a_norm(c1) / a_norm(c2));
    const float seg = radius * radius / 2 * (angle - glibc_sinf(angle));
    const double tri = 0.5 * a_norm(a_cross(a_sub(q[0], ip), a_sub(q[1], ip)));
    area = (float)(area - (seg + tri));
  }
  return full - area;
}

// rayGeometryDisk.hpp:266-354 for one disk (disk = {x,y,z,r}, nrm = its normal)
VR_HD float disk_exposed_area(const AreaParams &p, const float *disk, const float *nrm) {
  const int dirs[2] = {p.firstDir, p.secondDir};
  const int bcs[2] = {p.bcFirst, p.bcSecond};
  if (p.D == 3) {
    float a = (float)(disk[3] * disk[3] * VR_PI);
    if (bcs[0] == 2 && bcs[1] == 2) // IGNORE on both: no wall clips a disk
      return a;
    if (dirs[0] != 2 && dirs[1] != 2)
      return disk_area_inside_xy(disk, nrm, p.minC[0], p.minC[1], p.maxC[0], p.maxC[1]);
    const double eps = 1e-3;
    for (int s = 0; s < 2; ++s) {
      const float v = disk[dirs[s]];
      const float dlo = v - p.minC[dirs[s]], dhi = v - p.maxC[dirs[s]];
      if ((dlo < 0 ? -dlo : dlo) < eps || (dhi < 0 ? -dhi : dhi) < eps)
        a /= 2;
    }
    return a;
  }
  float a = 2 * disk[3];
  const int ax = dirs[0];
  for (int side = 0; side < 2; ++side) {
    const float wallc = side ? p.maxC[ax] : p.minC[ax];
    float dist = disk[ax] - wallc;
    dist = dist < 0 ? -dist : dist;
    if (bcs[0] != 2 && dist < disk[3]) {
      float t = 1 - nrm[ax] * nrm[ax];
      if (t > 1e-4) {
        t = dist / __builtin_sqrtf(t);
        if (t < disk[3])
          a -= disk[3] - t;
      }
    }
  }
  return a;
}

} // namespace vr
