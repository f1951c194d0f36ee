// Host-side mirror of the reference's trust-region layer (src/ParOptTrustRegion.h:15-493,
// src/ParOptCompactEigenvalueApprox.h:9-206): the subproblem interface, the quadratic and
// compact-eigenvalue subproblems, the infeasibility (steering) problem and the SL1QP driver with the
// adaptive penalty update.  Same method names and argument meaning; every n-sized operation is a
// launch of the kernels the interior-point path already uses (mdot / panel_axpy / panel_lincomb).
#pragma once
#include <string>
#include <vector>

#include "ip.hpp"

namespace po {

int k_tr_bounds(Ctx *c, const double *xk, const double *lb, const double *ub, double tr, int64_t n,
                double *lk, double *uk);
int k_kkt_error(Ctx *c, const double *xk, const double *lb, const double *ub, const double *t,
                double relax, int64_t n, double out[2]);

// c(s) = c0 + g0^T s + 1/2 s^T H M H^T s   (ParOptCompactEigenApprox, .cpp:23-120); M, Minv row-major
class CompactEigenApprox {
 public:
  CompactEigenApprox(Ctx *ctx, int64_t n, int N);
  ~CompactEigenApprox();
  int allocate();
  std::vector<const double *> hPointers() const;
  Ctx *ctx;
  int64_t n;
  int N;
  double c0;
  Vec *g0;
  std::vector<double> M, Minv;
  std::vector<Vec *> hvecs;
  std::vector<po_vec> hhandles;
};

// B = B_qn - z0 * H M H^T as one compact matrix over [Z_qn | H]  (ParOptEigenQuasiNewton :122-291)
class EigenQuasiNewton : public CompactQuasiNewton {
 public:
  EigenQuasiNewton(CompactQuasiNewton *qn, CompactEigenApprox *eigh, int index);
  void reset() override;
  int update(Vec *s, Vec *y, int *rc) override;
  int updateMult(Vec *x, const double *z, Vec *zw) override;
  int mult(Vec *x, Vec *y) override;
  int multAdd(double alpha, Vec *x, Vec *y) override;
  int getCompactMat(double *b0_, const double **d0_, const double **M_, Vec ***Z_) override;
  int getMaxLimitedMemorySize() override;
  void applyCompactInverse(double *rz) const override;
  int size() const override;
  double diag() const override;
  std::vector<const double *> zPointers() const override;

  CompactQuasiNewton *qn;  // borrowed, may be null
  CompactEigenApprox *eigh;
  int index;
  double z0;
  int use_qn_objective;

 private:
  int qnSize() const { return (qn && use_qn_objective) ? qn->size() : 0; }
  std::vector<double> dall, Mall;
  std::vector<Vec *> Zall;
};

class TrustRegionSubproblem : public Problem {
 public:
  TrustRegionSubproblem(Problem *p);
  virtual ~TrustRegionSubproblem();
  int allocate();
  virtual CompactQuasiNewton *getQuasiNewton() = 0;
  virtual int initModelAndBounds(double tr_size);
  virtual int setTrustRegionBounds(double tr_size);
  virtual int evalTrialStepAndUpdate(int update_flag, Vec *step, const double *z, Vec *zw, double *fobj,
                                     double *cons) = 0;
  virtual int acceptTrialStep(Vec *step, const double *z, Vec *zw) = 0;
  virtual void rejectTrialStep();
  virtual int getQuasiNewtonUpdateType() const { return qn_update_type; }
  // linear model values at a step: f = fk + gk.s, cons = ck + Ak s (one panel-dot pass)
  int evalLinearModel(Vec *step, double *f, double *cons);

  // ParOptProblem side, seen by the interior-point solver
  int getVarsAndBounds(Vec *x, Vec *l, Vec *u) override;
  int evalSparseCon(Vec *step, Vec *out) override;
  int addSparseJacobian(double alpha, Vec *x, Vec *px, Vec *out) override;
  int addSparseJacobianTranspose(double alpha, Vec *x, Vec *pzw, Vec *out) override;
  int setSparseJacobianTranspose(double alpha, Vec *, Vec *pzw, Vec *out) override {
    return prob->setSparseJacobianTranspose(alpha, xk, pzw, out);
  }
  int addSparseInnerProduct(double alpha, Vec *x, Vec *cvec, Vec *A) override;
  int sparseJacobianPanel(Vec *x, Vec *d, const double *const *P, int nv, double *const *U,
                          Vec *work) override;
  int sparseApplyK0(Vec *, Vec *d, Vec *cw, const double *bx, const double *bw, Vec *yx, Vec *yw,
                    Vec *wwork) override {
    return prob->sparseApplyK0(xk, d, cw, bx, bw, yx, yw, wwork);
  }
  int sparseFactor(Vec *, Vec *d, Vec *cw) override { return prob->sparseFactor(xk, d, cw); }
  bool sparseGramGroups(Vec *, GramGroups *g) override { return prob->sparseGramGroups(xk, g); }
  bool sparseTransposeColumn(double alpha, Vec *, Vec *pzw, GroupCol *col) override {
    return prob->sparseTransposeColumn(alpha, xk, pzw, col);
  }
  int sparseFactorFromSlacks(Vec *, Vec *d, const WVars &v, Vec *cw) override {
    return prob->sparseFactorFromSlacks(xk, d, v, cw);
  }
  int setSparseJacobian(double alpha, Vec *, Vec *px, Vec *out) override {
    return prob->setSparseJacobian(alpha, xk, px, out);
  }
  int sparseHalfSolve(double *const *U, int nv, Vec *cw, const double **weights) override {
    return prob->sparseHalfSolve(U, nv, cw, weights);
  }
  const char *sparseFactorInfo() override { return prob->sparseFactorInfo(); }
  long sparseFactorBreakdowns() override { return prob->sparseFactorBreakdowns(); }
  int sparseCorrection(const double *const *U, int nv, const double *alpha, Vec *cw, Vec *out, Vec *acc) override {
    return prob->sparseCorrection(U, nv, alpha, cw, out, acc);
  }
  int writeOutput(int iter, Vec *x) override { return prob->writeOutput(iter, x); }
  // the model evaluations issue their reductions through the internal launchers and finish with
  // BatchScope::end_then: inside the interior point's line-search batch they share its collective + host sync
  bool reductionsBatchable() override { return prob->reductionsBatchable(); }

  Problem *prob;
  int m;
  Vec *xk, *lk, *uk, *lb, *ub, *gk, *gt, *t, *xtemp;
  std::vector<Vec *> Ak, At;
  double fk, ft;
  std::vector<double> ck, ct;

 protected:
  int evalTrialPoint(Vec *step, double *fobj, double *cons);
  int lagrangianGradientDifference(const double *z, Vec *zw);  // into t (:187-205)
  void acceptModel();
  int qn_update_type;
  // landing areas of the model reductions (the values may arrive at the flush of an enclosing batch)
  std::vector<double> eo_dots, lin_dots;
  // Z^T step of the last evalObjCon(step): evalObjConGradient at the same step buffer (the accepted trial point)
  // does not stream the panel again.  Dropped whenever the model or the quasi-Newton panel changes.
  std::vector<double> zts_cache;
  const double *zts_ptr = nullptr;
  bool zts_valid = false;
};

class QuadraticSubproblem : public TrustRegionSubproblem {  // :27-466
 public:
  QuadraticSubproblem(Problem *p, CompactQuasiNewton *qn_) : TrustRegionSubproblem(p), qn(qn_) {
    linear_constraints = 1;  // c(s) = ck + Ak s: the Jacobian is constant within a subproblem solve
  }
  CompactQuasiNewton *getQuasiNewton() override { return qn; }
  int evalTrialStepAndUpdate(int update_flag, Vec *step, const double *z, Vec *zw, double *fobj,
                             double *cons) override;
  int acceptTrialStep(Vec *step, const double *z, Vec *zw) override;
  int evalObjCon(Vec *step, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *step, Vec *g, Vec **Ac) override;
  CompactQuasiNewton *qn;
};

typedef int (*EigenModelUpdate)(void *user, Vec *x, CompactEigenApprox *approx);

class EigenSubproblem : public TrustRegionSubproblem {  // ParOptCompactEigenvalueApprox.cpp:295-724
 public:
  EigenSubproblem(Problem *p, EigenQuasiNewton *approx_)
      : TrustRegionSubproblem(p), approx(approx_), update_model(nullptr), update_user(nullptr) {}
  CompactQuasiNewton *getQuasiNewton() override { return approx; }
  int initModelAndBounds(double tr_size) override;
  int evalTrialStepAndUpdate(int update_flag, Vec *step, const double *z, Vec *zw, double *fobj,
                             double *cons) override;
  int acceptTrialStep(Vec *step, const double *z, Vec *zw) override;
  int evalObjCon(Vec *step, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *step, Vec *g, Vec **Ac) override;
  // every constraint model but the eigenvalue one is linear in the step
  const std::vector<char> *constantJacobianMask() override {
    const_mask.assign(m, 1);
    if (approx->index >= 0 && approx->index < m) const_mask[approx->index] = 0;
    return &const_mask;
  }
  std::vector<char> const_mask;
  EigenQuasiNewton *approx;
  EigenModelUpdate update_model;
  void *update_user;

 private:
  int modelDots(Vec *step, std::vector<double> &dots, int *kq);
};


// A subproblem written by the USER (round 5): the seven virtuals of ParOptTrustRegionSubproblem
// (src/ParOptTrustRegion.h:15-151) and the ParOptProblem side the interior point solves, as a callback table
// (po_trsub_create_callbacks; the facade's ParOptTrustRegionSubproblem subclasses and Python's TrustRegionSubproblem
// arrive here).  The driver and the steering problem read the linear model through the base-class members (xk, fk, gk,
// ck, Ak, lb, ub): they are BORROWED from the user's getLinearModel after every call that moves the base point.
class CallbackSubproblem : public TrustRegionSubproblem {
 public:
  CallbackSubproblem(Problem *p, const po_trsub_callbacks &cb_) : TrustRegionSubproblem(p), cb(cb_) {}
  ~CallbackSubproblem();
  int allocateModel();  // no storage of its own: borrows the user's model vectors
  CompactQuasiNewton *getQuasiNewton() override;
  int initModelAndBounds(double tr_size) override;
  int setTrustRegionBounds(double tr_size) override;
  int evalTrialStepAndUpdate(int update_flag, Vec *step, const double *z, Vec *zw, double *fobj,
                             double *cons) override;
  int acceptTrialStep(Vec *step, const double *z, Vec *zw) override;
  void rejectTrialStep() override;
  int getQuasiNewtonUpdateType() const override;
  int getVarsAndBounds(Vec *x, Vec *l, Vec *u) override;
  int evalObjCon(Vec *step, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *step, Vec *g, Vec **Ac) override;
  bool reductionsBatchable() override { return false; }  // user code: every reduction returns its value at once
  // sparse_constraints_are_model: `prob` already IS the model's problem side (called with the step)
  int evalSparseCon(Vec *step, Vec *out) override {
    return cb.sparse_constraints_are_model ? prob->evalSparseCon(step, out) : TrustRegionSubproblem::evalSparseCon(step, out);
  }
  int addSparseJacobian(double alpha, Vec *x, Vec *px, Vec *out) override {
    return cb.sparse_constraints_are_model ? prob->addSparseJacobian(alpha, x, px, out)
                                           : TrustRegionSubproblem::addSparseJacobian(alpha, x, px, out);
  }
  int addSparseJacobianTranspose(double alpha, Vec *x, Vec *pzw, Vec *out) override {
    return cb.sparse_constraints_are_model ? prob->addSparseJacobianTranspose(alpha, x, pzw, out)
                                           : TrustRegionSubproblem::addSparseJacobianTranspose(alpha, x, pzw, out);
  }
  int addSparseInnerProduct(double alpha, Vec *x, Vec *cvec, Vec *A) override {
    return cb.sparse_constraints_are_model ? prob->addSparseInnerProduct(alpha, x, cvec, A)
                                           : TrustRegionSubproblem::addSparseInnerProduct(alpha, x, cvec, A);
  }
  po_trsub_callbacks cb;

  // re-reads getLinearModel (called by the driver's own entry points; po_trsub_sync_linear_model for a model the
  // user changed outside the driver)
  int syncLinearModel();

 private:
  void dropModel();
};

class InfeasSubproblem : public Problem {  // :468-650
 public:
  enum { CONSTANT_OBJECTIVE = 0, LINEAR_OBJECTIVE = 1, SUBPROBLEM_OBJECTIVE = 2 };
  enum { LINEAR_CONSTRAINT = 0, SUBPROBLEM_CONSTRAINT = 1 };
  InfeasSubproblem(TrustRegionSubproblem *sub_, int objective_, int constraint_);
  int getVarsAndBounds(Vec *x, Vec *l, Vec *u) override { return sub->getVarsAndBounds(x, l, u); }
  int evalObjCon(Vec *step, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *step, Vec *g, Vec **Ac) override;
  bool reductionsBatchable() override { return sub->reductionsBatchable(); }
  const std::vector<char> *constantJacobianMask() override {
    return constraint == SUBPROBLEM_CONSTRAINT ? sub->constantJacobianMask() : nullptr;
  }
  std::vector<double> eo_cs, eo_cl;  // landing areas (see TrustRegionSubproblem::eo_dots)
  double eo_fs = 0.0, eo_fl = 0.0;
  int evalSparseCon(Vec *step, Vec *out) override { return sub->evalSparseCon(step, out); }
  int addSparseJacobian(double a, Vec *x, Vec *px, Vec *out) override {
    return sub->addSparseJacobian(a, x, px, out);
  }
  int addSparseJacobianTranspose(double a, Vec *x, Vec *pzw, Vec *out) override {
    return sub->addSparseJacobianTranspose(a, x, pzw, out);
  }
  int setSparseJacobianTranspose(double a, Vec *x, Vec *pzw, Vec *out) override {
    return sub->setSparseJacobianTranspose(a, x, pzw, out);
  }
  int addSparseInnerProduct(double a, Vec *x, Vec *cvec, Vec *A) override {
    return sub->addSparseInnerProduct(a, x, cvec, A);
  }
  int sparseJacobianPanel(Vec *x, Vec *d, const double *const *P, int nv, double *const *U,
                          Vec *work) override {
    return sub->sparseJacobianPanel(x, d, P, nv, U, work);
  }
  int sparseApplyK0(Vec *x, Vec *d, Vec *cw, const double *bx, const double *bw, Vec *yx, Vec *yw,
                    Vec *wwork) override {
    return sub->sparseApplyK0(x, d, cw, bx, bw, yx, yw, wwork);
  }
  int sparseFactor(Vec *x, Vec *d, Vec *cw) override { return sub->sparseFactor(x, d, cw); }
  bool sparseGramGroups(Vec *x, GramGroups *g) override { return sub->sparseGramGroups(x, g); }
  bool sparseTransposeColumn(double alpha, Vec *x, Vec *pzw, GroupCol *col) override {
    return sub->sparseTransposeColumn(alpha, x, pzw, col);
  }
  int sparseFactorFromSlacks(Vec *x, Vec *d, const WVars &v, Vec *cw) override {
    return sub->sparseFactorFromSlacks(x, d, v, cw);
  }
  int setSparseJacobian(double a, Vec *x, Vec *px, Vec *out) override { return sub->setSparseJacobian(a, x, px, out); }
  int sparseHalfSolve(double *const *U, int nv, Vec *cw, const double **weights) override {
    return sub->sparseHalfSolve(U, nv, cw, weights);
  }
  const char *sparseFactorInfo() override { return sub->sparseFactorInfo(); }
  long sparseFactorBreakdowns() override { return sub->sparseFactorBreakdowns(); }
  int sparseCorrection(const double *const *U, int nv, const double *alpha, Vec *cw, Vec *out, Vec *acc) override {
    return sub->sparseCorrection(U, nv, alpha, cw, out, acc);
  }
  TrustRegionSubproblem *sub;
  int objective, constraint;
  double obj_scale;
};

typedef int (*TrIterationFn)(void *user, int iter);

class TrustRegion {
 public:
  // everything assembled from the options, as ParOptOptimizer does for algorithm = "tr"
  explicit TrustRegion(Problem *prob);
  // ParOptTrustRegion(subproblem, options) (src/ParOptTrustRegion.cpp:660-718): the caller owns the subproblem
  // (and, through it, the quasi-Newton object); the interior-point solver arrives with optimize(ip)
  explicit TrustRegion(TrustRegionSubproblem *subproblem);
  ~TrustRegion();
  Options &options() { return ip ? ip->options : opts; }
  // compact eigenvalue model for constraint `index` with N curvature directions (before optimize)
  int setEigenModel(int N, int index, EigenModelUpdate update, void *user);
  // ext != nullptr: ParOptTrustRegion::optimize(ParOptInteriorPoint*) (.cpp:2365-2384) -- the caller's solver, built
  // on the subproblem, is used (borrowed); the trust-region options set on this object are carried into its registry
  int optimize(InteriorPoint *ext = nullptr);
  int build();  // quasi-Newton, subproblem, interior point from the current options (idempotent)
  int initialize();  // ParOptTrustRegion::initialize (.cpp:1086-1099)
  void setPenaltyGamma(double gamma);           // .cpp:1049-1055
  void setPenaltyGammaArray(const double *g);   // .cpp:1062-1068

  Problem *prob;
  Ctx *ctx;
  Options opts;  // registry until the interior-point object exists, then ip->options is the one
  CompactQuasiNewton *qn;
  CompactEigenApprox *eigh;
  EigenQuasiNewton *eqn;
  TrustRegionSubproblem *sub;
  InfeasSubproblem *infeas;
  InteriorPoint *ip;
  po_qn_s qn_handle;

  int m, nineq;
  std::vector<double> penalty_gamma;
  double tr_size;
  int iter_count, subproblem_iters, adaptive_subproblem_iters;
  std::string history;
  TrIterationFn iter_cb;
  void *iter_cb_user;
  // last iteration's table row (for observers): fobj, infeas, l1, linfty, smax, tr, rho, model_reduc,
  // zav, zmax, gav, gmax
  double row[12];
  std::string row_info;
  // last line of the interior point's iteration table in the two subproblem solves of the latest iteration
  // ([0] steering / restoration solve, [1] the QP): how each solve ended (po_tr_get_last_solve_lines)
  std::string last_solve_line[2];
  void captureSolveLine(int which);

 private:
  bool own_sub = true, own_ip = true, state_ready = false;
  int initState();  // penalty parameters and radius from the options (the reference's constructor)
  int eig_N, eig_index;
  EigenModelUpdate eig_update;
  void *eig_user;
  Vec *tvec;
  double infeasOf(const double *c, const double *weights) const;
  int computeKKTError(const double *z, Vec *zw, double *l1, double *linfty);
  int minimizeInfeas(std::vector<double> *best_con_infeas);
  int sl1qpOptimize();
  int filterOptimize();
  // filter set :896-966
  struct FilterElement {
    double f, h;
  };
  std::vector<FilterElement> filter;
  int acceptableByPair(double f_new, double h_new, double f_old, double h_old);
  int acceptableByFilter(double f, double h);
  void addToFilter(double f, double h);
  void appendRow(const double vals[12], const std::string &info, double seconds);
  int sl1qpUpdate(Vec *step, const double *z, Vec *zw, double *infeas, double *l1, double *linfty);
  void flushHistory();
};

}  // namespace po

struct po_tr_s {
  po::TrustRegion *tr;
  void *eig_holder;  // capi_tr.cpp: storage behind the eigenvalue-model callbacks, owned by the handle
};
struct po_eig_s {
  po::CompactEigenApprox *e;
};
