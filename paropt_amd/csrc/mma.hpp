// Host-side mirror of ParOptMMA (reference src/ParOptMMA.h:22-192, src/ParOptMMA.cpp): the method of
// moving asymptotes.  The object is BOTH the separable rational subproblem -- a ParOptProblem handed
// to the interior-point solver, with its diagonal Hessian (use_diag_hessian = 1, use_line_search = 0,
// .cpp:344-346) -- and the outer driver.  Every n-sized operation is an element-wise kernel (mma.hip)
// or one of the reductions of the interior-point path; the subproblem's objective / constraint values
// are two mdot passes over the coefficient panels [p0 | p_i] and [q0 | q_i].
#pragma once
#include <string>
#include <vector>

#include "ip.hpp"

namespace po {

struct MmaParams {
  double movlim, init_off, contract, relax, min_off, max_off, eps, delta;
};
int k_mma_asymptotes(Ctx *c, const double *x, const double *x1, const double *x2, const double *lb,
                     const double *ub, const MmaParams &p, int first, int64_t n, double *L, double *U);
int k_mma_coef(Ctx *c, const double *x, const double *lb, const double *ub, const double *L, const double *U,
               const double *g, const MmaParams &p, int64_t n, double *alpha, double *beta, double *p0,
               double *q0);
// pi = (U-x)^2 max(0,-A), qi = (x-L)^2 max(0,A) ; *bsum = sum pi/(U-x) + qi/(x-L)
int k_mma_pq(Ctx *c, const double *x, const double *L, const double *U, const double *A, int64_t n, double *pi,
             double *qi, double *bsum);
int k_mma_inv(Ctx *c, const double *x, const double *L, const double *U, int64_t n, double *uinv, double *linv);
// out_0 = ui^2 P_0 - li^2 Q_0 ; out_j = li^2 Q_j - ui^2 P_j (j >= 1)
int k_mma_grad(Ctx *c, const double *x, const double *L, const double *U, const double *const *P,
               const double *const *Q, int nv, int64_t n, double *const *out);
// h = 2 sum_j w_j (ui^3 P_j + li^3 Q_j)
int k_mma_hdiag(Ctx *c, const double *x, const double *L, const double *U, const double *const *P,
                const double *const *Q, const double *w, int nv, int64_t n, double *h);

typedef int (*MmaIterationFn)(void *user, int iter);

class MMA : public Problem {
 public:
  explicit MMA(Problem *prob);
  ~MMA();
  Options &options() { return ip ? ip->options : opts; }
  int build();
  int optimize();

  // ParOptProblem side (:795-1052)
  int getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) override;
  int evalObjCon(Vec *x, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *x, Vec *g, Vec **Ac) override;
  int evalHvecProduct(Vec *x, const double *z, Vec *zw, Vec *px, Vec *hvec) override;
  int evalHessianDiag(Vec *x, const double *z, Vec *zw, Vec *hdiag) override;
  int evalSparseCon(Vec *x, Vec *out) override;
  int addSparseJacobian(double alpha, Vec *x, Vec *px, Vec *out) override;
  int addSparseJacobianTranspose(double alpha, Vec *x, Vec *pzw, Vec *out) override;
  int addSparseInnerProduct(double alpha, Vec *x, Vec *cvec, Vec *A) override;
  int sparseJacobianPanel(Vec *x, Vec *d, const double *const *P, int nv, double *const *U,
                          Vec *work) override;
  int sparseApplyK0(Vec *, Vec *d, Vec *cw, const double *bx, const double *bw, Vec *yx, Vec *yw,
                    Vec *wwork) override {
    return prob->sparseApplyK0(xvec, d, cw, bx, bw, yx, yw, wwork);
  }
  int sparseFactor(Vec *, Vec *d, Vec *cw) override { return prob->sparseFactor(xvec, d, cw); }
  int sparseHalfSolve(double *const *U, int nv, Vec *cw, const double **weights) override {
    return prob->sparseHalfSolve(U, nv, cw, weights);
  }
  const char *sparseFactorInfo() override { return prob->sparseFactorInfo(); }
  long sparseFactorBreakdowns() override { return prob->sparseFactorBreakdowns(); }
  int sparseCorrection(const double *const *U, int nv, const double *alpha, Vec *cw, Vec *out, Vec *acc) override {
    return prob->sparseCorrection(U, nv, alpha, cw, out, acc);
  }

  Problem *prob;
  Options opts;
  InteriorPoint *ip;
  int m;
  int use_true_mma, mma_iter, subproblem_iter;
  Vec *xvec, *x1vec, *x2vec, *lbvec, *ubvec, *gvec, *Lvec, *Uvec, *alphavec, *betavec, *p0vec, *q0vec, *rvec,
      *zlvec, *zuvec, *uinv, *linv, *cwvec, *zwvec;
  std::vector<Vec *> Avecs, pivecs, qivecs;
  double fobj;
  std::vector<double> cons, b, z;
  std::string history;
  MmaIterationFn iter_cb;
  void *iter_cb_user;
  double last_row[5];  // fobj, l1, linfty, l1_lambda, infeas of the last table row

 private:
  MmaParams params() ;
  int allocate();
  int initializeSubProblem(Vec *xv);
  int computeKKTError(double *l1, double *linfty, double *infeas);
  void setMultipliers();
  void flushHistory();
};

}  // namespace po

struct po_mma_s {
  po::MMA *mma;
};
