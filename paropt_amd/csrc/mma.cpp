// Method of moving asymptotes on the device (see mma.hpp); control flow of ParOptMMA::optimize /
// initializeSubProblem / computeKKTError (reference src/ParOptMMA.cpp:318-757), quirks included.
#include "mma.hpp"

#include <math.h>
#include <stdio.h>

#include <algorithm>

#include "tr.hpp"

namespace po {

MMA::MMA(Problem *p)
    : Problem(p->ctx, p->nlocal, p->ncon, p->ninequality), prob(p), ip(nullptr), m(p->ncon), use_true_mma(1),
      mma_iter(0), subproblem_iter(0), xvec(nullptr), x1vec(nullptr), x2vec(nullptr), lbvec(nullptr),
      ubvec(nullptr), gvec(nullptr), Lvec(nullptr), Uvec(nullptr), alphavec(nullptr), betavec(nullptr),
      p0vec(nullptr), q0vec(nullptr), rvec(nullptr), zlvec(nullptr), zuvec(nullptr), uinv(nullptr),
      linv(nullptr), cwvec(nullptr), zwvec(nullptr), fobj(0.0), cons(p->ncon, 0.0), b(p->ncon, 0.0),
      z(p->ncon, 0.0), iter_cb(nullptr), iter_cb_user(nullptr) {
  offset = p->offset;
  nglobal = p->nglobal;
  nwcon = p->nwcon;
  nwinequality = p->nwinequality;
  opts.addMMADefaults();
  opts.set("output_file", "");
  for (double &v : last_row) v = 0.0;
}

MMA::~MMA() {
  delete ip;
  Vec *all[] = {xvec, x1vec, x2vec, lbvec, ubvec, gvec, Lvec, Uvec, alphavec, betavec, p0vec,
                q0vec, rvec, zlvec, zuvec, uinv, linv, cwvec, zwvec};
  for (Vec *v : all) vec_decref(v);
  for (Vec *v : Avecs) vec_decref(v);
  for (Vec *v : pivecs) vec_decref(v);
  for (Vec *v : qivecs) vec_decref(v);
}

int MMA::allocate() {  // initialize() :131-232
  const int64_t n = nlocal;
  Vec **all[] = {&xvec, &x1vec, &x2vec, &lbvec, &ubvec, &gvec, &Lvec, &Uvec, &alphavec, &betavec,
                 &p0vec, &q0vec, &rvec, &zlvec, &zuvec, &uinv, &linv};
  for (Vec **v : all) {
    *v = vec_new(ctx, n);
    if (!*v) return PO_ERR_HIP;
  }
  for (int i = 0; i < m; i++) {
    Vec *a = vec_new(ctx, n), *pv = vec_new(ctx, n), *qv = vec_new(ctx, n);
    if (!a || !pv || !qv) return PO_ERR_HIP;
    Avecs.push_back(a);
    pivecs.push_back(pv);
    qivecs.push_back(qv);
  }
  if (nwcon > 0) {
    cwvec = vec_new(ctx, nwcon);
    zwvec = vec_new(ctx, nwcon);
    if (!cwvec || !zwvec) return PO_ERR_HIP;
  }
  PO_TRY(k_fill(ctx, alphavec->d, n, 0.0));
  PO_TRY(k_fill(ctx, betavec->d, n, 1.0));
  if (prob->getVarsAndBounds(xvec, lbvec, ubvec) != 0) return PO_ERR_USER;
  if (!prob->useUpperBounds()) PO_TRY(k_fill(ctx, ubvec->d, n, 10.0));
  if (!prob->useLowerBounds()) PO_TRY(k_fill(ctx, lbvec->d, n, -9.0));
  return PO_OK;
}

int MMA::build() {
  if (ip) return PO_OK;
  if (m + 1 > kMaxPanel) {
    set_error("MMA: %d constraints exceed the panel width %d", m, kMaxPanel - 1);
    return PO_ERR_ARG;
  }
  PO_TRY(allocate());
  ip = new InteriorPoint(this);
  ip->options = opts;
  PO_TRY(ip->allocate());
  return PO_OK;
}

MmaParams MMA::params() {
  Options &o = options();
  MmaParams p;
  p.movlim = o.real("mma_move_limit");
  p.init_off = o.real("mma_init_asymptote_offset");
  p.contract = o.real("mma_asymptote_contract");
  p.relax = o.real("mma_asymptote_relax");
  p.min_off = o.real("mma_min_asymptote_offset");
  p.max_off = o.real("mma_max_asymptote_offset");
  p.eps = o.real("mma_eps_regularization");
  p.delta = o.real("mma_delta_regularization");
  return p;
}

int MMA::computeKKTError(double *l1, double *linfty, double *infeas) {  // :406-484
  const double relax = options().real("mma_bound_relax");
  const int64_t n = nlocal;
  std::vector<const double *> P;
  std::vector<double> cf;
  for (int i = 0; i < m; i++) {
    P.push_back(Avecs[i]->d);
    cf.push_back(-z[i]);
  }
  if (relax <= 0.0) {
    P.push_back(zlvec->d);
    cf.push_back(-1.0);
    P.push_back(zuvec->d);
    cf.push_back(1.0);
  }
  PO_TRY(k_panel_axpy(ctx, rvec->d, 1.0, gvec->d, 0.0, cf.data(), P.data(), (int)P.size(), n));
  if (nwcon > 0) {
    if (prob->addSparseJacobianTranspose(-1.0, xvec, zwvec, rvec) != 0) return PO_ERR_USER;
  }
  if (relax <= 0.0) {
    PO_TRY(k_reduce1(ctx, RED_ASUM, rvec->d, nullptr, n, l1));
    PO_TRY(k_reduce1(ctx, RED_AMAX, rvec->d, nullptr, n, linfty));
  } else {
    double out[2];
    PO_TRY(k_kkt_error(ctx, xvec->d, lbvec->d, ubvec->d, rvec->d, relax, n, out));
    *l1 = out[0];
    *linfty = out[1];
  }
  *infeas = 0.0;
  for (int i = 0; i < m; i++) *infeas += fabs(std::min(0.0, cons[i]));
  return PO_OK;
}

int MMA::initializeSubProblem(Vec *xv) {  // :523-757
  const int64_t n = nlocal;
  PO_TRY(k_copy(ctx, x2vec->d, x1vec->d, n));
  PO_TRY(k_copy(ctx, x1vec->d, xvec->d, n));
  if (xv && xv != xvec) PO_TRY(k_copy(ctx, xvec->d, xv->d, n));
  if (prob->evalObjCon(xvec, &fobj, cons.data()) != 0) {
    fprintf(stderr, "ParOptMMA: Objective evaluation failed\n");
    return PO_ERR_USER;
  }
  if (prob->evalObjConGradient(xvec, gvec, Avecs.data()) != 0) {
    fprintf(stderr, "ParOptMMA: Gradient evaluation failed\n");
    return PO_ERR_USER;
  }
  if (nwcon > 0 && prob->evalSparseCon(xvec, cwvec) != 0) return PO_ERR_USER;
  {  // the table row :571-596
    double l1 = 0.0, linfty = 0.0, infeas = 0.0, l1_lambda = 0.0;
    PO_TRY(computeKKTError(&l1, &linfty, &infeas));
    for (int i = 0; i < m; i++) l1_lambda += fabs(z[i]);
    const double vals[5] = {fobj, l1, linfty, l1_lambda, infeas};
    for (int i = 0; i < 5; i++) last_row[i] = vals[i];
    if (ctx->rank == 0) {
      char line[256];
      if (mma_iter % 10 == 0) {
        snprintf(line, sizeof(line), "\n%5s %8s %15s %9s %9s %9s %9s\n", "MMA", "sub-iter", "fobj", "l1-opt",
                 "linft-opt", "l1-lambd", "infeas");
        history += line;
      }
      snprintf(line, sizeof(line), "%5d %8d %15.6e %9.3e %9.3e %9.3e %9.3e\n", mma_iter, subproblem_iter, fobj,
               l1, linfty, l1_lambda, infeas);
      history += line;
    }
    if (iter_cb) iter_cb(iter_cb_user, mma_iter);
  }
  const MmaParams p = params();
  PO_TRY(k_mma_asymptotes(ctx, xvec->d, x1vec->d, x2vec->d, lbvec->d, ubvec->d, p, mma_iter < 2 ? 1 : 0, n,
                          Lvec->d, Uvec->d));
  PO_TRY(k_mma_coef(ctx, xvec->d, lbvec->d, ubvec->d, Lvec->d, Uvec->d, gvec->d, p, n, alphavec->d,
                    betavec->d, p0vec->d, q0vec->d));
  if (use_true_mma) {
    for (int i = 0; i < m; i++) {
      double bs = 0.0;
      PO_TRY(k_mma_pq(ctx, xvec->d, Lvec->d, Uvec->d, Avecs[i]->d, n, pivecs[i]->d, qivecs[i]->d, &bs));
      b[i] = -(cons[i] + bs);
    }
  }
  mma_iter++;
  return PO_OK;
}

void MMA::setMultipliers() {  // :384-400
  Vec *x = nullptr, *zl = nullptr, *zu = nullptr;
  const double *zz = nullptr;
  ip->getOptimizedPoint(&x, &zz, &zl, &zu);
  for (int i = 0; i < m; i++) z[i] = zz[i];
  Vec *wv[5];
  ip->getOptimizedSparse(wv);
  if (wv[0] && zwvec) k_copy(ctx, zwvec->d, wv[0]->d, nwcon);
  if (zl) k_copy(ctx, zlvec->d, zl->d, nlocal);
  if (zu) k_copy(ctx, zuvec->d, zu->d, nlocal);
}

int MMA::optimize() {  // :318-379
  PO_TRY(build());
  Options &o = ip->options;
  const int max_it = o.integer("mma_max_iterations");
  const double infeas_tol = o.real("mma_infeas_tol"), l1_tol = o.real("mma_l1_tol"),
               linfty_tol = o.real("mma_linfty_tol");
  use_true_mma = o.integer("mma_use_constraint_linearization") ? 0 : 1;
  PO_TRY(o.set("use_diag_hessian", 1));
  PO_TRY(o.set("use_line_search", 0));
  history.clear();
  PO_TRY(initializeSubProblem(xvec));
  PO_TRY(ip->resetDesignAndBounds());
  for (int i = 0; i < max_it; i++) {
    int rc = ip->optimize(nullptr);
    if (rc != 0 && rc != 1) return rc;
    setMultipliers();
    Vec *x = nullptr;
    ip->getOptimizedPoint(&x, nullptr, nullptr, nullptr);
    PO_TRY(initializeSubProblem(x));
    PO_TRY(ip->resetDesignAndBounds());
    // the reference calls computeKKTError(&infeas, &l1, &linfty) on a function declared as
    // (l1, linfty, infeas) (:364-366): the names below therefore hold permuted quantities
    double infeas = 0.0, l1 = 0.0, linfty = 0.0;
    PO_TRY(computeKKTError(&infeas, &l1, &linfty));
    if (infeas < infeas_tol && (l1 < l1_tol || linfty < linfty_tol)) break;
  }
  flushHistory();
  return 0;
}

void MMA::flushHistory() {
  const std::string fname = options().str("mma_output_file");
  if (ctx->rank != 0 || fname.empty()) return;
  FILE *fp = fopen(fname.c_str(), "w");
  if (!fp) return;
  fputs("ParOptMMA (paropt_amd, MI355X)\n", fp);
  fputs(history.c_str(), fp);
  fclose(fp);
}

// ---- the subproblem ---------------------------------------------------------------------------------
int MMA::getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) {  // :795-799
  if (!xvec) {  // before build(): the interior point's constructor probe
    k_fill(ctx, x->d, nlocal, 0.5);
    k_fill(ctx, lb->d, nlocal, 0.0);
    k_fill(ctx, ub->d, nlocal, 1.0);
    return 0;
  }
  PO_TRY(k_copy(ctx, x->d, xvec->d, nlocal));
  PO_TRY(k_copy(ctx, lb->d, alphavec->d, nlocal));
  PO_TRY(k_copy(ctx, ub->d, betavec->d, nlocal));
  return 0;
}

int MMA::evalObjCon(Vec *xv, double *fval, double *cvals) {  // :804-866
  const int64_t n = nlocal;
  if (k_mma_inv(ctx, xv->d, Lvec->d, Uvec->d, n, uinv->d, linv->d) != PO_OK) return 1;
  const int nv = use_true_mma ? m + 1 : 1;
  std::vector<const double *> P, Q;
  P.push_back(p0vec->d);
  Q.push_back(q0vec->d);
  if (use_true_mma) {
    for (int i = 0; i < m; i++) {
      P.push_back(pivecs[i]->d);
      Q.push_back(qivecs[i]->d);
    }
  }
  std::vector<double> du(nv, 0.0), dl(nv, 0.0);
  if (k_mdot(ctx, uinv->d, P.data(), nv, n, du.data()) != PO_OK) return 1;
  if (k_mdot(ctx, linv->d, Q.data(), nv, n, dl.data()) != PO_OK) return 1;
  *fval = du[0] + dl[0];
  if (use_true_mma) {
    for (int i = 0; i < m; i++) cvals[i] = -((du[1 + i] + dl[1 + i]) + b[i]);
  } else if (m > 0) {
    // linearised constraints: cons + A (x - x0)
    const double mone[1] = {-1.0};
    const double *vv[1] = {xvec->d};
    if (k_panel_axpy(ctx, rvec->d, 1.0, xv->d, 0.0, mone, vv, 1, n) != PO_OK) return 1;
    std::vector<const double *> A;
    for (Vec *a : Avecs) A.push_back(a->d);
    if (k_mdot(ctx, rvec->d, A.data(), m, n, cvals) != PO_OK) return 1;
    for (int i = 0; i < m; i++) cvals[i] += cons[i];
  }
  return 0;
}

int MMA::evalObjConGradient(Vec *xv, Vec *gv, Vec **Ac) {  // :871-924
  subproblem_iter++;
  const int64_t n = nlocal;
  std::vector<const double *> P, Q;
  std::vector<double *> out;
  P.push_back(p0vec->d);
  Q.push_back(q0vec->d);
  out.push_back(gv->d);
  if (use_true_mma) {
    for (int i = 0; i < m; i++) {
      P.push_back(pivecs[i]->d);
      Q.push_back(qivecs[i]->d);
      out.push_back(Ac[i]->d);
    }
  }
  if (k_mma_grad(ctx, xv->d, Lvec->d, Uvec->d, P.data(), Q.data(), (int)P.size(), n, out.data()) != PO_OK) return 1;
  if (!use_true_mma && m > 0) {
    std::vector<double *> dst;
    std::vector<const double *> src;
    for (int i = 0; i < m; i++) {
      dst.push_back(Ac[i]->d);
      src.push_back(Avecs[i]->d);
    }
    if (k_panel_lincomb(ctx, dst.data(), 1.0, src.data(), 0.0, nullptr, m, n) != PO_OK) return 1;
  }
  return 0;
}

int MMA::evalHvecProduct(Vec *xv, const double *, Vec *, Vec *px, Vec *hvec) {  // :929-962 (objective only)
  const double one[1] = {1.0};
  const double *P[1] = {p0vec->d}, *Q[1] = {q0vec->d};
  if (k_mma_hdiag(ctx, xv->d, Lvec->d, Uvec->d, P, Q, one, 1, nlocal, hvec->d) != PO_OK) return 1;
  return k_mul(ctx, hvec->d, 1.0, hvec->d, px->d, nlocal) == PO_OK ? 0 : 1;
}

int MMA::evalHessianDiag(Vec *xv, const double *zz, Vec *, Vec *hdiag) {  // :967-1010
  std::vector<const double *> P, Q;
  std::vector<double> w;
  P.push_back(p0vec->d);
  Q.push_back(q0vec->d);
  w.push_back(1.0);
  if (use_true_mma) {
    for (int i = 0; i < m; i++) {
      P.push_back(pivecs[i]->d);
      Q.push_back(qivecs[i]->d);
      w.push_back(zz[i]);
    }
  }
  return k_mma_hdiag(ctx, xv->d, Lvec->d, Uvec->d, P.data(), Q.data(), w.data(), (int)P.size(), nlocal,
                     hdiag->d) == PO_OK
             ? 0
             : 1;
}

int MMA::evalSparseCon(Vec *x, Vec *out) {  // :1015-1021
  if (nwcon <= 0) return 0;
  if (k_copy(ctx, out->d, cwvec->d, nwcon) != PO_OK) return 1;
  if (prob->addSparseJacobian(1.0, xvec, x, out) != 0) return 1;
  return prob->addSparseJacobian(-1.0, xvec, xvec, out);
}
int MMA::addSparseJacobian(double alpha, Vec *, Vec *px, Vec *out) {
  return nwcon > 0 ? prob->addSparseJacobian(alpha, xvec, px, out) : 0;
}
int MMA::addSparseJacobianTranspose(double alpha, Vec *, Vec *pzw, Vec *out) {
  return nwcon > 0 ? prob->addSparseJacobianTranspose(alpha, xvec, pzw, out) : 0;
}
int MMA::addSparseInnerProduct(double alpha, Vec *, Vec *cvec, Vec *A) {
  return nwcon > 0 ? prob->addSparseInnerProduct(alpha, xvec, cvec, A) : 0;
}
int MMA::sparseJacobianPanel(Vec *, Vec *d, const double *const *P, int nv, double *const *U, Vec *work) {
  return prob->sparseJacobianPanel(xvec, d, P, nv, U, work);
}

}  // namespace po
