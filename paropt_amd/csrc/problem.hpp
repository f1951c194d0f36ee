// Host-side mirror of the reference's ParOptProblem interface (src/ParOptProblem.h:42-296) for
// dense constraints (nwcon = 0): same method names and argument meaning, device vectors.
#pragma once
#include <vector>

#include "core.hpp"
#include "qn.hpp"

namespace po {

class Problem {
 public:
  Problem(Ctx *c, int64_t nlocal_, int ncon_, int nineq_)
      : ctx(c), nlocal(nlocal_), offset(0), nglobal(nlocal_), ncon(ncon_), ninequality(nineq_) {}
  virtual ~Problem() {}
  virtual int getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) = 0;
  virtual int evalObjCon(Vec *x, double *fobj, double *cons) = 0;
  virtual int evalObjConGradient(Vec *x, Vec *g, Vec **Ac) = 0;
  virtual int computeQuasiNewtonUpdateCorrection(Vec *x, const double *z, Vec *s, Vec *y) { return 0; }
  virtual int writeOutput(int iter, Vec *x) { return 0; }
  virtual int useLowerBounds() { return 1; }
  virtual int useUpperBounds() { return 1; }

  Ctx *ctx;
  int64_t nlocal, offset, nglobal;
  int ncon, ninequality;
};

// C callback table (the shape of the reference's Cython trampolines, src/CyParOptProblem.h:44-69)
class CallbackProblem : public Problem {
 public:
  CallbackProblem(Ctx *c, int64_t nlocal, int ncon, int nineq, const po_problem_callbacks &cb_)
      : Problem(c, nlocal, ncon, nineq), cb(cb_) {}
  int getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) override;
  int evalObjCon(Vec *x, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *x, Vec *g, Vec **Ac) override;
  int computeQuasiNewtonUpdateCorrection(Vec *x, const double *z, Vec *s, Vec *y) override;
  int writeOutput(int iter, Vec *x) override;
  po_problem_callbacks cb;
};

// Device-resident separable workloads (DESIGN.md "Workloads").
class SeparableProblem : public Problem {
 public:
  SeparableProblem(Ctx *c, int kind, int64_t nglobal, int ncon, uint64_t seed, double eig_min,
                   double eig_max);
  ~SeparableProblem();
  int init();
  int getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) override;
  int evalObjCon(Vec *x, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *x, Vec *g, Vec **Ac) override;

  int kind;
  uint64_t seed;
  double eig_min, eig_max;
  Vec *q, *b;             // objective data
  std::vector<Vec *> A;   // constraint rows a_j (device)
  std::vector<double> beta;
};

}  // namespace po

struct po_problem_s {
  po::Problem *p;
};
