// The separable random quadratic of the trust-region parity goldens (tests/golden/tr_*; oracle/ref_driver.cpp SepProblem
// "quadratic") as a USER problem on include/ParOptAMD.hpp, host arrays through getArray like the reference's examples:
//   f(x) = sum 1/2 q_i x_i^2 + b_i x_i,  c_j(x) = beta_j + a_j . x >= 0,  -5 <= x <= 5,
// every array a pure function of (seed, array id, index).  Shared by examples/eigenvalue_amd.cpp and
// examples/user_subproblem_amd.cpp.
#pragma once
#include <stdint.h>

#include "ParOptAMD.hpp"

static uint64_t splitmix64(uint64_t z) {
  z += 0x9E3779B97F4A7C15ULL;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}
// counter hash of the synthetic workloads: u01(seed, array id, index) in [0, 1)
static double u01(uint64_t seed, uint64_t aid, uint64_t i) {
  const uint64_t h = splitmix64(seed * 0x9E3779B97F4A7C15ULL + aid * 0xD1B54A32D192ED03ULL + i);
  return (double)(h >> 11) * (1.0 / 9007199254740992.0);
}

class SeparableQuadratic : public ParOptProblem {
 public:
  SeparableQuadratic(po_ctx ctx, int n, int m, uint64_t _seed) : ParOptProblem(ctx), seed(_seed) {
    setProblemSizes(n, m, 0);
    setNumInequalities(m, 0);
  }
  void getVarsAndBounds(ParOptVec *xvec, ParOptVec *lbvec, ParOptVec *ubvec) {
    ParOptScalar *x, *lb, *ub;
    xvec->getArray(&x);
    lbvec->getArray(&lb);
    ubvec->getArray(&ub);
    for (int i = 0; i < nvars; i++) {
      x[i] = -2.0 + u01(seed, 3, i);
      lb[i] = -5.0;
      ub[i] = 5.0;
    }
  }
  int evalObjCon(ParOptVec *xvec, ParOptScalar *fobj, ParOptScalar *cons) {
    ParOptScalar *x;
    xvec->getArray(&x);
    double f = 0.0;
    for (int i = 0; i < nvars; i++) {
      const double q = 1.0 + 99.0 * u01(seed, 1, i), b = u01(seed, 2, i);
      f += 0.5 * q * x[i] * x[i] + b * x[i];
    }
    *fobj = f;
    for (int j = 0; j < ncon; j++) {
      double s = 0.0;
      for (int i = 0; i < nvars; i++) s += u01(seed, 100 + j, i) * x[i];
      cons[j] = s + u01(seed, 4, j);
    }
    return 0;
  }
  int evalObjConGradient(ParOptVec *xvec, ParOptVec *gvec, ParOptVec **Ac) {
    ParOptScalar *x, *g;
    xvec->getArray(&x);
    gvec->getArray(&g);
    for (int i = 0; i < nvars; i++) g[i] = (1.0 + 99.0 * u01(seed, 1, i)) * x[i] + u01(seed, 2, i);
    for (int j = 0; j < ncon; j++) {
      ParOptScalar *a;
      Ac[j]->getArray(&a);
      for (int i = 0; i < nvars; i++) a[i] = u01(seed, 100 + j, i);
    }
    return 0;
  }
  uint64_t seed;
};

