// vr_particles.hpp — the device-side particle registry.
//
// The reference's extension point is AbstractParticle (rayParticle.hpp:21-81): virtual
// surfaceCollision / surfaceReflection / initNew called per hit on the host, each handed the primitive id, the
// material id and `const TracingData *globalData`.  Its own GPU path replaces the virtuals by a table of device
// callables per particle (COLLISION, REFLECTION, INIT: gpu/raygCallableConfig.hpp:7-18,
// gpu/pipelines/Particle.cuh:15-39).  Here a particle MODEL is a struct of __device__ functions with one fixed
// shape (below); models are compiled into trace_kernel — the two built-ins as their own instantiations (no
// run-time dispatch in the hot kernels), everything else through the EXTENDED instantiations, which select the
// model of TraceParams::particleKind from the type list `Registry`.
//
// Adding a model: write the struct, append it to `Registry` (its position is its kind id, VR_PARTICLE_* in
// include/viennaray_amd.h), give the host class a deviceModel() (include/viennaray_amd/viennaray.hpp).  A model
// reads its own parameters from ModelCtx::params (vr_particle::params, 8 floats) and the caller's global data
// (vr_set_global_data) from ModelCtx::global — no change to the kernels or the C ABI.
//
//   sticking : the first member of surfaceReflection's result (rayParticle.hpp:44-50): the share of the weight
//              that stays on primitive `primID`.  `base` is the particle's stickingProbability after the
//              per-material map (gpu::Particle::materialSticking)
//   reflect  : the second member: the direction after the hit; consumes engine outputs exactly like the host code
//   collide  : surfaceCollision (rayParticle.hpp:60-68): what one hit adds to the particle's data labels; called
//              for the closest disk and for every overlapping neighbour with that disk's own normal and id
//              (rayTraceKernel.hpp:284-300); credit(label, value) adds to this primitive's entry of a label
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

// ---------------------------------------------------------------------------------------------------------
// what a model sees of the launch: its own parameters and the caller's global data (Trace::setGlobalData,
// rayTrace.hpp:137-145: a borrowed, read-only TracingData handed to every surfaceCollision / surfaceReflection)
// ---------------------------------------------------------------------------------------------------------
struct GlobalData {
  const float *vec;        // [numVec][stride], indexed by the ORIGINAL primitive id (the primID of the host callbacks)
  const float *scalars;    // [numScalars]
  unsigned numVec, stride, numScalars;
  // TracingData::getVectorData(v)[i]; a vector the caller did not provide reads as 0
  __device__ __forceinline__ float vector(unsigned v, unsigned i) const {
    return (v < numVec && i < stride) ? vec[(size_t)v * stride + i] : 0.f;
  }
  __device__ __forceinline__ float scalar(unsigned s) const { return s < numScalars ? scalars[s] : 0.f; }
};

struct ModelCtx {
  const float *params; // vr_particle::params
  GlobalData global;
};

__device__ __forceinline__ ModelCtx model_ctx(const TraceParams &p) {
  ModelCtx m;
  m.params = p.particleParams;
  m.global.vec = p.globalVec;
  m.global.scalars = p.globalScalars;
  m.global.numVec = p.numGlobalVec;
  m.global.stride = p.globalStride;
  m.global.numScalars = p.numGlobalScalars;
  return m;
}

// ---------------------------------------------------------------------------------------------------------
// the models
// ---------------------------------------------------------------------------------------------------------
// DiffuseParticle (rayParticle.hpp:126-163)
struct ModelDiffuse {
  static constexpr int kNumData = 1;
  static constexpr bool kNeedsFull = false;
  __device__ static float sticking(const ModelCtx &, unsigned, float base) { return base; }
  template <int D>
  __device__ static V3 reflect(const ModelCtx &, const V3 &, const V3 &n, Rng &rng, unsigned &t2) {
    return reflection_diffuse<D>(n, rng, t2);
  }
  template <class Credit>
  __device__ static void collide(const ModelCtx &, float w, const V3 &, const V3 &, unsigned, Credit &&credit) {
    credit(0, w);
  }
};

// SpecularParticle (rayParticle.hpp:165-204)
struct ModelSpecular : ModelDiffuse {
  template <int D>
  __device__ static V3 reflect(const ModelCtx &, const V3 &rayDir, const V3 &n, Rng &, unsigned &) {
    return reflect_specular(rayDir, n);
  }
};

// surfaceReflection = ReflectionConedCosine(coneAngle) (rayReflection.hpp:52-120); params[0] = maxConeAngle
struct ModelConedCosine : ModelDiffuse {
  static constexpr bool kNeedsFull = true; // (double-precision trigonometry: the instantiation with the rare options)
  template <int D>
  __device__ static V3 reflect(const ModelCtx &m, const V3 &rayDir, const V3 &n, Rng &rng, unsigned &t2) {
    return reflection_coned_cosine<D>(rayDir, n, rng, t2, m.params[0]);
  }
};

// a DiffuseParticle with TWO data labels: label 0 += w, label 1 += w * max(0, -d.n)
struct ModelDiffuseCosine : ModelDiffuse {
  static constexpr int kNumData = 2;
  template <class Credit>
  __device__ static void collide(const ModelCtx &, float w, const V3 &rayDir, const V3 &n, unsigned, Credit &&credit) {
    credit(0, w);
    const float cosTheta = -vdot(rayDir, n);
    credit(1, w * fmaxf(cosTheta, 0.f));
  }
};

// The ViennaPS pattern the global data exists for: a diffuse particle whose sticking falls with the coverage of
// the surface it meets, sticking = s0 * (1 - coverage[primID]), coverage = global vector params[0] (default 0).
// Host form: surfaceReflection returns {s0 * (1 - globalData->getVectorData(v)[primID]), ReflectionDiffuse}.
struct ModelCoverageSticking : ModelDiffuse {
  __device__ static float sticking(const ModelCtx &m, unsigned primID, float base) {
    return base * (1.f - m.global.vector((unsigned)m.params[0], primID));
  }
};

// ---------------------------------------------------------------------------------------------------------
// registry: position = kind id (VR_PARTICLE_*)
// ---------------------------------------------------------------------------------------------------------
template <class... M> struct ModelList {
  static constexpr int size = (int)sizeof...(M);
};
#ifdef VR_USER_MODEL_FILE
// a model registered at run time (vr_register_particle_model): the caller's source defines `struct VrUserModel` with the
// shape above — typically derived from one of the models here — and becomes the last entry of this module's registry
#include VR_USER_MODEL_FILE
using Registry = ModelList<ModelDiffuse, ModelSpecular, ModelConedCosine, ModelDiffuseCosine, ModelCoverageSticking, VrUserModel>;
#else
using Registry = ModelList<ModelDiffuse, ModelSpecular, ModelConedCosine, ModelDiffuseCosine, ModelCoverageSticking>;
#endif
constexpr int VR_BUILTIN_MODELS = 5; // (a run-time model is number VR_BUILTIN_MODELS inside its own code object)

template <int I, class List> struct ModelAt;
template <int I, class M0, class... M> struct ModelAt<I, ModelList<M0, M...>> : ModelAt<I - 1, ModelList<M...>> {};
template <class M0, class... M> struct ModelAt<0, ModelList<M0, M...>> {
  using type = M0;
};

struct Particles {
  static constexpr int count = Registry::size;

  // run f(Model{}) for the model of `kind`; FULL: the instantiation that carries the models with kNeedsFull
  // (the host never sends such a kind to the lean one)
  template <bool FULL, int I = 0, class F> __device__ __forceinline__ static void with(int kind, F &&f) {
    if constexpr (I < Registry::size) {
      using M = typename ModelAt<I, Registry>::type;
      if constexpr (FULL || !M::kNeedsFull) {
        if (kind == I) {
          f(M{});
          return;
        }
      }
      with<FULL, I + 1>(kind, static_cast<F &&>(f));
    }
  }

  template <int I = 0> __host__ __device__ static int numData(int kind) {
    if constexpr (I < Registry::size) {
      return kind == I ? ModelAt<I, Registry>::type::kNumData : numData<I + 1>(kind);
    } else {
      return 1;
    }
  }
  template <int I = 0> __host__ __device__ static bool needsFull(int kind) {
    if constexpr (I < Registry::size) {
      return kind == I ? ModelAt<I, Registry>::type::kNeedsFull : needsFull<I + 1>(kind);
    } else {
      return false;
    }
  }

  template <bool FULL>
  __device__ __forceinline__ static float sticking(int kind, const ModelCtx &m, unsigned primID, float base) {
    float s = base;
    with<FULL>(kind, [&](auto model) { s = decltype(model)::sticking(m, primID, base); });
    return s;
  }
  template <int D, bool FULL>
  __device__ __forceinline__ static V3 reflect(int kind, const ModelCtx &m, const V3 &rayDir, const V3 &n, Rng &rng,
                                               unsigned &t2) {
    V3 r = rayDir;
    with<FULL>(kind, [&](auto model) { r = decltype(model)::template reflect<D>(m, rayDir, n, rng, t2); });
    return r;
  }
  template <bool FULL, class Credit>
  __device__ __forceinline__ static void collide(int kind, const ModelCtx &m, float w, const V3 &rayDir, const V3 &n,
                                                 unsigned primID, Credit &&credit) {
    with<FULL>(kind, [&](auto model) { decltype(model)::collide(m, w, rayDir, n, primID, credit); });
  }
};

} // namespace vr
