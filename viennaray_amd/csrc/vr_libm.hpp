// vr_libm.hpp — sincosf / powf exactly as the reference's libm computes them.
//
// The reference samples the source direction with glibc's float `sincosf` and `powf`
// (raySourceRandom.hpp:77-78 via rayUtil.hpp:247-256).  To trace bit-identical rays the
// device evaluates the same algorithm: the double-precision polynomial kernels of glibc
// 2.35's flt-32 `sincosf` and `powf` (from the Arm Optimized Routines), with the
// multiply-adds fused exactly where the x86-64 FMA variants (`__sincosf_fma`,
// `__powf_fma`, selected by IFUNC on every AVX2+FMA CPU) fuse them.  The tables below are
// the published constants of those routines.  tests/aux/libm_check.cpp verifies the host
// build of this header against the running glibc over every float in the ranges used.
//
// Only the argument ranges the tracer needs are handled: sincosf for |x| < 120,
// powf for 0 <= x < 2, 0 < y <= 1 (finite, normal).
#pragma once
#include <cstdint>
#include <cstring>

#if defined(__HIPCC__) || defined(__HIP__)
#include <hip/hip_runtime.h>
#define VR_HD __host__ __device__ __forceinline__
#else
#define VR_HD inline
#endif

namespace vr {

VR_HD uint32_t vr_asuint(float f) {
  uint32_t u;
  memcpy(&u, &f, 4);
  return u;
}
VR_HD float vr_asfloat(uint32_t u) {
  float f;
  memcpy(&f, &u, 4);
  return f;
}
VR_HD uint64_t vr_asuint64(double d) {
  uint64_t u;
  memcpy(&u, &d, 8);
  return u;
}
VR_HD double vr_asdouble(uint64_t u) {
  double d;
  memcpy(&d, &u, 8);
  return d;
}

// glibc 2.35 sysdeps/ieee754/flt-32/s_sincosf.[ch] (x86-64 fma variant)
VR_HD void glibc_sincosf(float y, float &sinp, float &cosp) {
  const double c0 = 0x1p0, c1 = -0x1.ffffffd0c621cp-2, c2 = 0x1.55553e1068f19p-5, c3 = -0x1.6c087e89a359dp-10,
               c4 = 0x1.99343027bf8c3p-16;
  const double s1 = -0x1.555545995a603p-3, s2 = 0x1.1107605230bc4p-7, s3 = -0x1.994eb3774cf24p-13;
  const double hpi_inv = 0x1.45f306dc9c883p+23, hpi = 0x1.921fb54442d18p+0;
  const double x = (double)y;
  const uint32_t top = (vr_asuint(y) >> 20) & 0x7ffu;
  double xs, x2, flip = 1.0;
  int n = 0;
  if (top <= 0x3f3u) { // |y| < pi/4
    if (top <= 0x397u) { // |y| < 2^-12
      sinp = y;
      cosp = 1.0f;
      return;
    }
    xs = x;
    x2 = x * x;
  } else { // pi/4 <= |y| < 120: reduce_fast
    const double r = x * hpi_inv;
    n = ((int32_t)r + 0x800000) >> 24;
    const double xr = __builtin_fma(-(double)n, hpi, x);
    const double sgn = ((n & 3) == 1 || (n & 3) == 2) ? -1.0 : 1.0; // sign[n & 3] = {1,-1,-1,1}
    xs = xr * sgn;
    x2 = xr * xr;
    if (n & 2)
      flip = -1.0; // second table: cosine coefficients negated
  }
  const double C0 = flip * c0, C1 = flip * c1, C2 = flip * c2, C3 = flip * c3, C4 = flip * c4;
  const double s1v = __builtin_fma(x2, s3, s2);
  const double c2v = __builtin_fma(x2, C4, C3);
  const double x3 = x2 * xs;
  const double x4 = x2 * x2;
  const double x5 = x2 * x3;
  const double x6 = x2 * x4;
  const double c1v = __builtin_fma(x2, C1, C0);
  const double s = __builtin_fma(x3, s1, xs);
  const double c = __builtin_fma(x4, C2, c1v);
  const float sv = (float)__builtin_fma(s1v, x5, s);
  const float cv = (float)__builtin_fma(c2v, x6, c);
  if (n & 1) {
    sinp = cv;
    cosp = sv;
  } else {
    sinp = sv;
    cosp = cv;
  }
}

// glibc 2.35 sysdeps/ieee754/flt-32/e_powf.c (x86-64 fma variant), main path
VR_HD float glibc_powf(float x, float y) {
  if (x == 0.f)
    return 0.f; // y > 0
  if (x == 1.f)
    return 1.f;
  static const double invc[16] = {0x1.661ec79f8f3bep+0, 0x1.571ed4aaf883dp+0, 0x1.49539f0f010b0p+0,
                                  0x1.3c995b0b80385p+0, 0x1.30d190c8864a5p+0, 0x1.25e227b0b8ea0p+0,
                                  0x1.1bb4a4a1a343fp+0, 0x1.12358f08ae5bap+0, 0x1.0953f419900a7p+0,
                                  0x1.0000000000000p+0, 0x1.e608cfd9a47acp-1, 0x1.ca4b31f026aa0p-1,
                                  0x1.b2036576afce6p-1, 0x1.9c2d163a1aa2dp-1, 0x1.886e6037841edp-1,
                                  0x1.767dcf5534862p-1};
  static const double logc[16] = {-0x1.efec65b963019p-2, -0x1.b0b6832d4fca4p-2, -0x1.7418b0a1fb77bp-2,
                                  -0x1.39de91a6dcf7bp-2, -0x1.01d9bf3f2b631p-2, -0x1.97c1d1b3b7af0p-3,
                                  -0x1.2f9e393af3c9fp-3, -0x1.960cbbf788d5cp-4, -0x1.a6f9db6475fcep-5,
                                  0x0.0p+0,              0x1.338ca9f24f53dp-4,  0x1.476a9543891bap-3,
                                  0x1.e840b4ac4e4d2p-3,  0x1.40645f0c6651cp-2,  0x1.88e9c2c1b9ff8p-2,
                                  0x1.ce0a44eb17bccp-2};
  static const uint64_t exp2tab[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
      0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
      0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
      0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
      0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  const double A0 = 0x1.27616c9496e0bp-2, A1 = -0x1.71969a075c67ap-2, A2 = 0x1.ec70a6ca7baddp-2,
               A3 = -0x1.7154748bef6c8p-1, A4 = 0x1.71547652ab82bp+0;
  const double SHIFT = 0x1.8p+47, C0 = 0x1.c6af84b912394p-5, C1 = 0x1.ebfce50fac4f3p-3, C2 = 0x1.62e42ff0c52d6p-1;
  // log2_inline
  const uint32_t ix = vr_asuint(x);
  const uint32_t tmp = ix - 0x3f330000u;
  const uint32_t i = (tmp >> 19) & 15u;
  const uint32_t top = tmp & 0xff800000u;
  const uint32_t iz = ix - top;
  const int k = (int32_t)top >> 23;
  const double z = (double)vr_asfloat(iz);
  const double r = __builtin_fma(z, invc[i], -1.0);
  const double y0 = logc[i] + (double)k;
  const double py = __builtin_fma(r, A0, A1);
  const double pp = __builtin_fma(r, A2, A3);
  const double r2 = r * r;
  double q = __builtin_fma(r, A4, y0);
  const double r4 = r2 * r2;
  q = __builtin_fma(r2, pp, q);
  const double logx = __builtin_fma(py, r4, q);
  const double ylogx = (double)y * logx;
  // exp2_inline (sign_bias = 0)
  double kd = ylogx + SHIFT;
  const uint64_t ki = vr_asuint64(kd);
  kd -= SHIFT;
  const double rr = ylogx - kd;
  uint64_t t = exp2tab[ki % 32u];
  t += ki << 47;
  const double sc = vr_asdouble(t);
  const double zz = __builtin_fma(rr, C0, C1);
  const double rr2 = rr * rr;
  double yy = __builtin_fma(rr, C2, 1.0);
  yy = __builtin_fma(zz, rr2, yy);
  return (float)(yy * sc);
}

// glibc 2.35 sinf (sysdeps/ieee754/flt-32/s_sinf.c) evaluates the same polynomials in the same
// order as the sine half of sincosf: one routine serves both
VR_HD float glibc_sinf(float y) {
  float s, c;
  glibc_sincosf(y, s, c);
  return s;
}

// glibc 2.35 acosf (sysdeps/ieee754/flt-32/e_acosf.c, the fdlibm float routine; no FMA variant is
// selected for it on x86-64).  Used by the disk-area intersector (std::acos on floats,
// rayDiskBoundingBoxIntersector.hpp); tests/aux/libm_check.cpp compares it with the running
// glibc on every float in [-1, 1].
VR_HD float glibc_acosf(float x) {
  const float one = 1.0000000000e+00f, pi = 3.1415925026e+00f, pio2_hi = 1.5707962513e+00f,
              pio2_lo = 7.5497894159e-08f, pS0 = 1.6666667163e-01f, pS1 = -3.2556581497e-01f,
              pS2 = 2.0121252537e-01f, pS3 = -4.0055535734e-02f, pS4 = 7.9153501429e-04f, pS5 = 3.4793309169e-05f,
              qS1 = -2.4033949375e+00f, qS2 = 2.0209457874e+00f, qS3 = -6.8828397989e-01f, qS4 = 7.7038154006e-02f;
  const int32_t hx = (int32_t)vr_asuint(x);
  const int32_t ix = hx & 0x7fffffff;
  if (ix == 0x3f800000) // |x| == 1
    return hx > 0 ? 0.0f : pi + 2.0f * pio2_lo;
  if (ix > 0x3f800000)
    return (x - x) / (x - x); // NaN
  if (ix < 0x3f000000) { // |x| < 0.5
    if (ix <= 0x32800000)
      return pio2_hi + pio2_lo;
    const float z = x * x;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float r = p / q;
    return pio2_hi - (x - (pio2_lo - x * r));
  }
  if (hx < 0) { // x < -0.5
    const float z = (one + x) * 0.5f;
    const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
    const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
    const float s = __builtin_sqrtf(z);
    const float r = p / q;
    const float w = r * s - pio2_lo;
    return pi - 2.0f * (s + w);
  }
  // x > 0.5
  const float z = (one - x) * 0.5f;
  const float s = __builtin_sqrtf(z);
  const float df = vr_asfloat(vr_asuint(s) & 0xfffff000u);
  const float c = (z - df * df) / (s + df);
  const float p = z * (pS0 + z * (pS1 + z * (pS2 + z * (pS3 + z * (pS4 + z * pS5)))));
  const float q = one + z * (qS1 + z * (qS2 + z * (qS3 + z * qS4)));
  const float r = p / q;
  const float w = r * s + c;
  return 2.0f * (df + w);
}

// glibc 2.35 expf (sysdeps/ieee754/flt-32/e_expf.c, x86-64 fma variant), for x <= 0: the
// mean-free-path scatter probability 1 - exp(-t / lambda) (rayTraceKernel.hpp:181-182)
VR_HD float glibc_expf(float x) {
  static const uint64_t exp2tab[32] = {
      0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
      0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
      0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
      0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
      0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
      0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
      0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
      0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull};
  if (!(x >= -0x1.9fe368p6f)) // below: underflow to +0 (NaN arguments do not occur here)
    return 0.0f;
  const double N = 32.0;
  const double InvLn2N = 0x1.71547652b82fep+0 * N, SHIFT = 0x1.8p+52;
  const double C0 = 0x1.c6af84b912394p-5 / N / N / N, C1 = 0x1.ebfce50fac4f3p-3 / N / N, C2 = 0x1.62e42ff0c52d6p-1 / N;
  const double xd = (double)x;
  const double z = InvLn2N * xd;
  double kd = z + SHIFT;
  const uint64_t ki = vr_asuint64(kd);
  kd -= SHIFT;
  const double r = z - kd;
  uint64_t t = exp2tab[ki % 32u];
  t += ki << 47;
  const double sc = vr_asdouble(t);
  const double zz = __builtin_fma(C0, r, C1);
  const double r2 = r * r;
  double y = __builtin_fma(C2, r, 1.0);
  y = __builtin_fma(zz, r2, y);
  return (float)(y * sc);
}

// ---------------------------------------------------------------------------
// double-precision sin / cos for |x| <= 8 (the coned-cosine reflection's three argument ranges: theta in [0, 1.6],
// pi/2 * s in [0, pi/2], phi in [0, 2 pi)): a two-part Cody-Waite reduction by pi/2 (|n| <= 5: exact products) and
// the classic Sun fdlibm kernel polynomials (k_sin.c / k_cos.c, Copyright (C) 1993 Sun Microsystems, freely
// usable with this notice), error < 1 ulp like the libm the reference links.  The device library's sin / cos
// carry the reduction for arguments up to 1e308 — register pressure and code the tracer never needs.  After the
// narrowing to float this and glibc's result differ with probability ~1e-8 per sample (the rounding boundary of
// the float falls between the two doubles), the same as with the device library before.
// ---------------------------------------------------------------------------
VR_HD void vr_sincos_small(double x, double &sn, double &cs) {
  const double invpio2 = 6.36619772367581382433e-01, pio2_1 = 1.57079632673412561417e+00, pio2_1t = 6.07710050650619224932e-11;
  const double fn = __builtin_rint(x * invpio2);
  const int n = (int)fn;
  const double r0 = x - fn * pio2_1; // (exact: pio2_1 has 33 significant bits, |fn| <= 5)
  const double w = fn * pio2_1t;
  const double y0 = r0 - w;          // reduced argument, |y0| <= pi/4 (+ rounding)
  const double y1 = (r0 - y0) - w;   // its tail
  const double z = y0 * y0;
  // k_sin
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03, S3 = -1.98412698298579493134e-04,
               S4 = 2.75573137070700676789e-06, S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  const double v = z * y0;
  const double rs = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  const double ks = y0 - ((z * (0.5 * y1 - v * rs) - y1) - v * S1);
  // k_cos
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03, C3 = 2.48015872894767294178e-05,
               C4 = -2.75573143513906633035e-07, C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  const double rc = z * (C1 + z * (C2 + z * (C3 + z * (C4 + z * (C5 + z * C6)))));
  const double ay = y0 < 0. ? -y0 : y0;
  double kc;
  if (ay < 0.3) {
    kc = 1.0 - (0.5 * z - (z * rc - y0 * y1));
  } else {
    const double qx = ay > 0.78125 ? 0.28125 : ay * 0.25; // (fdlibm: x / 4 with the low word cleared; any qx near it serves)
    const double hz = 0.5 * z - qx;
    const double a = 1.0 - qx;
    kc = a - (hz - (z * rc - y0 * y1));
  }
  switch (n & 3) {
  case 0: sn = ks; cs = kc; break;
  case 1: sn = kc; cs = -ks; break;
  case 2: sn = -ks; cs = -kc; break;
  default: sn = -kc; cs = ks; break;
  }
}

} // namespace vr
