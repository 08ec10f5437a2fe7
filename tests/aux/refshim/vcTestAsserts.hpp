// Test-only stand-in for ViennaCore's vcTestAsserts.hpp (the three macros the reference's tests use), so
// that the reference's OWN test and example sources — read in place under /root/reference, never copied —
// can be syntax-checked against the drop-in façade (tests/test_facade_units.py).  Not shipped, not
// included by anything under include/ or viennaray_amd/.
#pragma once
#include <cmath>
#include <cstdio>
#include <cstdlib>

// (block statements: the reference's tests use the macros with and without a trailing semicolon)
#define VC_TEST_ASSERT(cond)                                                                                           \
  {                                                                                                                    \
    if (!(cond)) {                                                                                                     \
      std::fprintf(stderr, "assertion failed %s:%d: %s\n", __FILE__, __LINE__, #cond);                                 \
      std::abort();                                                                                                    \
    }                                                                                                                  \
  }
#define VC_TEST_ASSERT_ISCLOSE(a, b, eps) VC_TEST_ASSERT(std::fabs(double(a) - double(b)) <= double(eps))
#define VC_RUN_ALL_TESTS                                                                                               \
  viennacore::RunTest<double, 2>();                                                                                    \
  viennacore::RunTest<double, 3>();                                                                                    \
  viennacore::RunTest<float, 2>();                                                                                     \
  viennacore::RunTest<float, 3>();
