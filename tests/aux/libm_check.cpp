// libm_check — exhaustive comparison of vr_libm.hpp (host build) with the running glibc.
// usage: libm_check [quick]   prints mismatch counts; exit code 1 on any mismatch.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <omp.h>

#include "../../viennaray_amd/csrc/vr_libm.hpp"

int main(int argc, char **argv) {
  const bool quick = argc > 1;
  const uint32_t step = quick ? 97u : 1u;
  // sincosf on every float in [0, 2*pi] (the source's phi = (float)(2*pi*r1))
  const uint32_t hi = vr::vr_asuint(6.2831855f);
  long badS = 0, badC = 0, n = 0;
#pragma omp parallel for reduction(+ : badS, badC, n) schedule(static)
  for (long u = 0; u <= (long)hi; u += step) {
    const float x = vr::vr_asfloat((uint32_t)u);
    float s0, c0, s1, c1;
    sincosf(x, &s0, &c0);
    vr::glibc_sincosf(x, s1, c1);
    badS += std::memcmp(&s0, &s1, 4) != 0;
    badC += std::memcmp(&c0, &c1, 4) != 0;
    ++n;
  }
  std::printf("sincosf: %ld inputs, sin mismatches %ld, cos mismatches %ld\n", n, badS, badC);
  // powf(x, ee) on every float x in [0, 1] for the exponents ee = 1/(power+1)
  const float powers[] = {1.f, 2.f, 5.f, 20.f, 50.f, 100.f, 1000.f, 0.5f};
  long badP = 0;
  for (float pw : powers) {
    const float ee = 1.f / (pw + 1.f);
    long bad = 0, m = 0;
    const uint32_t one = vr::vr_asuint(1.0f);
#pragma omp parallel for reduction(+ : bad, m) schedule(static)
    for (long u = 0; u <= (long)one; u += step) {
      const float x = vr::vr_asfloat((uint32_t)u);
      if (u != 0 && u < 0x00800000)
        continue; // subnormals cannot come out of generate_canonical
      const float a = powf(x, ee), b = vr::glibc_powf(x, ee);
      bad += std::memcmp(&a, &b, 4) != 0;
      ++m;
    }
    std::printf("powf(x, %.9g): %ld inputs, mismatches %ld\n", ee, m, bad);
    badP += bad;
  }
  // acosf on every float in [-1, 1] and sinf on [0, 2*pi] (disk-area intersector)
  long badA = 0, nA = 0, badSf = 0;
  {
    const uint32_t one = vr::vr_asuint(1.0f);
#pragma omp parallel for reduction(+ : badA, nA) schedule(static)
    for (long u = 0; u <= (long)one; u += step) {
      for (int sg = 0; sg < 2; ++sg) {
        const float x = vr::vr_asfloat((uint32_t)u | (sg ? 0x80000000u : 0u));
        const float a = acosf(x), b = vr::glibc_acosf(x);
        badA += std::memcmp(&a, &b, 4) != 0;
        ++nA;
      }
    }
#pragma omp parallel for reduction(+ : badSf) schedule(static)
    for (long u = 0; u <= (long)hi; u += step) {
      const float x = vr::vr_asfloat((uint32_t)u);
      const float a = sinf(x), b = vr::glibc_sinf(x);
      badSf += std::memcmp(&a, &b, 4) != 0;
    }
  }
  std::printf("acosf: %ld inputs, mismatches %ld; sinf mismatches %ld\n", nA, badA, badSf);
  // expf on every float in [-110, 0] (mean-free-path scatter probability)
  long badE = 0, nE = 0;
  {
    const uint32_t top = vr::vr_asuint(110.0f);
#pragma omp parallel for reduction(+ : badE, nE) schedule(static)
    for (long u = 0; u <= (long)top; u += step) {
      const float x = -vr::vr_asfloat((uint32_t)u);
      const float a = expf(x), b = vr::glibc_expf(x);
      badE += std::memcmp(&a, &b, 4) != 0;
      ++nE;
    }
  }
  std::printf("expf: %ld inputs, mismatches %ld\n", nE, badE);
  // double-precision sin / cos of the coned-cosine reflection (vr_sincos_small, |x| <= 2 pi): within 1 ulp of the
  // running libm, and identical after the narrowing to float on the whole sample
  long badD = 0, badF = 0, nD = 0;
  {
    const long N = quick ? 4000000 : 400000000;
#pragma omp parallel for reduction(+ : badD, badF, nD) schedule(static)
    for (long i = 0; i <= N; ++i) {
      // a dense sweep of [0, 2 pi] plus the small-angle end
      const double x = (i & 1) ? 6.283185307179586 * (double)i / (double)N : 1.6 * (double)i / (double)N * (double)i / (double)N;
      double s, c;
      vr::vr_sincos_small(x, s, c);
      const double rs = std::sin(x), rc = std::cos(x);
      const double us = std::fabs(rs) > 0 ? std::nextafter(std::fabs(rs), 2.0) - std::fabs(rs) : 5e-324;
      const double uc = std::fabs(rc) > 0 ? std::nextafter(std::fabs(rc), 2.0) - std::fabs(rc) : 5e-324;
      badD += std::fabs(s - rs) > us || std::fabs(c - rc) > uc;
      badF += (float)s != (float)rs || (float)c != (float)rc;
      ++nD;
    }
  }
  std::printf("sincos (double, range-limited): %ld inputs, beyond 1 ulp %ld, float-narrowed mismatches %ld\n", nD, badD, badF);
  return (badS || badC || badP || badA || badSf || badE || badD || badF) ? 1 : 0;
}
