// vr_particles.hpp — the device-side particle registry.
//
// The reference's extension point is AbstractParticle (rayParticle.hpp:21-81): virtual
// surfaceCollision / surfaceReflection / initNew called per hit on the host.  Its own GPU
// path replaces the virtuals by a table of device callables per particle (COLLISION,
// REFLECTION, INIT: gpu/raygCallableConfig.hpp:7-18, gpu/pipelines/Particle.cuh:15-39).
// Here a particle model is a struct of __device__ functions; models are compiled into
// trace_kernel — the two built-ins as their own instantiations (no run-time dispatch in the
// hot kernels), everything else through the EXTENDED instantiation, which switches on
// TraceParams::particleKind.  Adding a model = one struct here + one case in the switch of
// `Particles::reflect` / `Particles::collide` + a VR_PARTICLE_* id in include/viennaray_amd.h
// (and the matching host class in include/viennaray_amd/viennaray.hpp).
//
//   collide : what one surface hit adds to the particle's data labels (TracingData vectors);
//             called for the closest disk and for every overlapping neighbour with that
//             disk's own normal (rayTraceKernel.hpp:284-300); label l of primitive q is
//             credit(l, q, value)
//   reflect : (sticking is resolved by the caller: per-material map or the particle's value)
//             the direction after the hit; consumes engine outputs exactly like the host code
#pragma once
#include "vr_device.hpp"

namespace vr {

// rayUtil.hpp:266-283 (Marsaglia)
__device__ __forceinline__ V3 pick_random_point_on_unit_sphere(Rng &rng, unsigned &t2) {
  float x, y;
  double x2py2;
  do {
    x = canon_f32(rng_next(rng, t2)) * 2.0f + -1.0f;
    y = canon_f32(rng_next(rng, t2)) * 2.0f + -1.0f;
    x2py2 = (double)(x * x + y * y);
  } while (x2py2 >= 1.);
  const double tmp = 2. * sqrt(1. - x2py2);
  x = (float)((double)x * tmp);
  y = (float)((double)y * tmp);
  const float z = (float)(1. - 2 * x2py2);
  return mk(x, y, z);
}

// rayReflection.hpp:31-50
template <int D> __device__ __forceinline__ V3 reflection_diffuse(const V3 &n, Rng &rng, unsigned &t2) {
  const V3 s = pick_random_point_on_unit_sphere(rng, t2);
  V3 r = mk(s.x + n.x, s.y + n.y, D == 3 ? s.z + n.z : 0.f);
  vnormalize(r);
  return r;
}

// rayReflection.hpp:52-120.  The accept-reject loop and the trigonometry run in double like the
// reference; sin / cos are vr_libm.hpp's range-limited pair (< 1 ulp, not glibc's bit for bit: after the
// narrowing to float the two can differ with probability ~1e-8 per sample).
template <int D>
__device__ __forceinline__ V3 reflection_coned_cosine(const V3 &rayDir, const V3 &n, Rng &rng, unsigned &t2,
                                                      float maxConeAngle) {
  if (maxConeAngle <= 0.f)
    return reflect_specular(rayDir, n);
  if ((double)maxConeAngle >= 1.57079632679489661923)
    return reflection_diffuse<D>(n, rng, t2);
  V3 w = reflect_specular(rayDir, n);
  vnormalize(w);
  V3 t, b;
  if (w.z < -0.999999f) {
    t = mk(0.f, -1.f, 0.f);
    b = mk(-1.f, 0.f, 0.f);
  } else {
    const float a = 1.f / (1.f + w.z);
    const float bx = -w.x * w.y * a, by = 1.f - w.y * w.y * a;
    t = mk(1.f - w.x * w.x * a, bx, -w.x);
    b = mk(bx, by, -w.y);
  }
  // (sin / cos: the range-limited pair of vr_libm.hpp — theta <= 1.6, pi/2 s <= pi/2, phi < 2 pi — three
  //  evaluations per reflection instead of six calls of the device library's full-range routines)
  double theta, sinTheta, cosTheta;
  for (;;) {
    const double u = sqrt(canon_f64(rng_next(rng, t2)));
    const double s = sqrt(fmax(1.0 - u, 0.0));
    theta = (double)maxConeAngle * s;
    double sinHalf, cosHalf;
    vr_sincos_small(1.57079632679489661923 * s, sinHalf, cosHalf);
    vr_sincos_small(theta, sinTheta, cosTheta);
    const double rhs = cosHalf * sinTheta;
    if (canon_f64(rng_next(rng, t2)) * theta * u <= rhs)
      break;
  }
  const float sinT = (float)sinTheta;
  const float cosT = (float)cosTheta;
  const double phi = 2.0 * 3.14159265358979323846 * canon_f64(rng_next(rng, t2));
  double sinPhi, cosPhi;
  vr_sincos_small(phi, sinPhi, cosPhi);
  const float sinP = (float)sinPhi, cosP = (float)cosPhi;
  V3 dir = mk(sinT * (cosP * t.x + sinP * b.x) + cosT * w.x, sinT * (cosP * t.y + sinP * b.y) + cosT * w.y,
              sinT * (cosP * t.z + sinP * b.z) + cosT * w.z);
  const float dp = vdot(dir, n);
  if (dp <= 0.f) {
    const float g = 2.f * dp;
    dir = mk(dir.x - g * n.x, dir.y - g * n.y, dir.z - g * n.z);
  }
  if (D == 2)
    dir.z = 0.f;
  vnormalize(dir);
  return dir;
}

struct Particles {
  // number of data labels of a kind
  __host__ __device__ static int numData(int kind) { return kind == P_DIFFUSE_COSINE ? 2 : 1; }

  // surfaceReflection: new direction (the engine outputs it draws are part of the contract)
  // CONED: the instantiation carries the coned-cosine model (P_EXT_FULL); the host never sends that kind to the other
  template <int D, bool CONED>
  __device__ __forceinline__ static V3 reflect(int kind, const TraceParams &p, const V3 &rayDir, const V3 &n, Rng &rng,
                                               unsigned &t2) {
    if (CONED && kind == P_CONED_COSINE)
      return reflection_coned_cosine<D>(rayDir, n, rng, t2, p.coneAngle);
    if (kind == P_SPECULAR)
      return reflect_specular(rayDir, n);
    return reflection_diffuse<D>(n, rng, t2); // P_DIFFUSE, P_DIFFUSE_COSINE
  }

  // surfaceCollision: `credit(label, value)` adds to this primitive's entry of a data label
  template <class Credit>
  __device__ __forceinline__ static void collide(int kind, float w, const V3 &rayDir, const V3 &n, Credit &&credit) {
    credit(0, w);
    if (kind == P_DIFFUSE_COSINE) {
      const float cosTheta = -vdot(rayDir, n);
      credit(1, w * fmaxf(cosTheta, 0.f));
    }
  }
};

} // namespace vr
