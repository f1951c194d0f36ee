// Trust-region driver and its subproblems on the device (see tr.hpp).  Control flow follows
// ParOptTrustRegion::sl1qpOptimize / sl1qpUpdate / minimizeInfeas / computeKKTError
// (reference src/ParOptTrustRegion.cpp:1105-1687, 2391-2472) decision by decision; the model
// evaluations are fused into single panel-dot passes:
//   f(s) = fk + gk.s + 1/2 (b0 s.s - (Z^T s)^T d M^-1 d (Z^T s)),  c_i(s) = ck_i + Ak_i.s
// is ONE mdot over [gk | Ak | Z | s] instead of qn->mult + c + 2 dots (:290-323).
#include "tr.hpp"

#include <math.h>
#include <stdio.h>
#include <string.h>

#include <algorithm>
#include <chrono>

namespace po {

static double now_seconds() {
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

// ================================================================================================
// compact eigenvalue approximation
// ================================================================================================
CompactEigenApprox::CompactEigenApprox(Ctx *ctx_, int64_t n_, int N_)
    : ctx(ctx_), n(n_), N(N_), c0(0.0), g0(nullptr), M((size_t)N_ * N_, 0.0), Minv((size_t)N_ * N_, 0.0) {}
CompactEigenApprox::~CompactEigenApprox() {
  vec_decref(g0);
  for (Vec *v : hvecs) vec_decref(v);
}
int CompactEigenApprox::allocate() {
  g0 = vec_new(ctx, n);
  if (!g0) return PO_ERR_HIP;
  for (int i = 0; i < N; i++) {
    Vec *v = vec_new(ctx, n);
    if (!v) return PO_ERR_HIP;
    hvecs.push_back(v);
    hhandles.push_back(static_cast<po_vec>(v));
  }
  return PO_OK;
}
std::vector<const double *> CompactEigenApprox::hPointers() const {
  std::vector<const double *> p;
  for (Vec *v : hvecs) p.push_back(v->d);
  return p;
}

EigenQuasiNewton::EigenQuasiNewton(CompactQuasiNewton *qn_, CompactEigenApprox *eigh_, int index_)
    : CompactQuasiNewton(eigh_->ctx, eigh_->n, 0, false), qn(qn_), eigh(eigh_), index(index_), z0(1.0),
      use_qn_objective(1) {}
void EigenQuasiNewton::reset() {
  if (qn) qn->reset();
}
int EigenQuasiNewton::update(Vec *, Vec *, int *rc) {  // :176-179
  if (rc) *rc = 0;
  return PO_OK;
}
int EigenQuasiNewton::updateMult(Vec *, const double *z, Vec *) {  // :181-187
  z0 = z[index];
  return 0;
}
int EigenQuasiNewton::getMaxLimitedMemorySize() {
  return eigh->N + (qn ? qn->getMaxLimitedMemorySize() : 0);
}
int EigenQuasiNewton::size() const { return qnSize() + eigh->N; }
double EigenQuasiNewton::diag() const { return (qn && use_qn_objective) ? qn->diag() : 0.0; }
std::vector<const double *> EigenQuasiNewton::zPointers() const {
  std::vector<const double *> p;
  if (qn && use_qn_objective) p = qn->zPointers();
  for (Vec *v : eigh->hvecs) p.push_back(v->d);
  return p;
}
// rz <- diag(d) M^-1 diag(d) rz with the block structure of :212-280: quasi-Newton block as is,
// eigenvalue block (z0inv Minv)^-1 = z0 M, applied with M itself as multAdd does (:52-64)
void EigenQuasiNewton::applyCompactInverse(double *rz) const {
  const int kq = qnSize(), N = eigh->N;
  if (kq > 0) qn->applyCompactInverse(rz);
  std::vector<double> out(N, 0.0);
  for (int i = 0; i < N; i++) {
    double s = 0.0;
    for (int j = 0; j < N; j++) s += eigh->M[(size_t)i * N + j] * rz[kq + j];
    out[i] = z0 * s;
  }
  for (int i = 0; i < N; i++) rz[kq + i] = out[i];
}
int EigenQuasiNewton::getCompactMat(double *b0_, const double **d0_, const double **M_, Vec ***Z_) {
  const int kq = qnSize(), N = eigh->N, k = kq + N;
  dall.assign(k, 1.0);
  Mall.assign((size_t)k * k, 0.0);
  Zall.clear();
  double b = 0.0;
  if (kq > 0 || (qn && use_qn_objective)) {
    const double *d0q, *Mq;
    Vec **Zq;
    qn->getCompactMat(&b, &d0q, &Mq, &Zq);
    for (int i = 0; i < kq; i++) {
      dall[i] = d0q[i];
      Zall.push_back(Zq[i]);
      for (int j = 0; j < kq; j++) Mall[i + (size_t)k * j] = Mq[i + (size_t)kq * j];
    }
  }
  const double z0inv = z0 != 0.0 ? 1.0 / z0 : 1.0;
  for (int i = 0; i < N; i++) {
    Zall.push_back(eigh->hvecs[i]);
    for (int j = 0; j < N; j++) Mall[(kq + i) + (size_t)k * (kq + j)] = z0inv * eigh->Minv[(size_t)i * N + j];
  }
  if (b0_) *b0_ = b;
  if (d0_) *d0_ = dall.data();
  if (M_) *M_ = Mall.data();
  if (Z_) *Z_ = Zall.data();
  return k;
}
int EigenQuasiNewton::multAdd(double alpha, Vec *x, Vec *y) {  // :198-204
  std::vector<const double *> zp = zPointers();
  const int k = (int)zp.size();
  std::vector<double> rz(k > 0 ? k : 1, 0.0);
  if (k > 0) {
    PO_TRY(k_mdot(ctx, x->d, zp.data(), k, n, rz.data()));
    applyCompactInverse(rz.data());
    for (int i = 0; i < k; i++) rz[i] = -alpha * rz[i];
  }
  return k_panel_axpy(ctx, y->d, alpha * diag(), x->d, 1.0, rz.data(), zp.data(), k, n);
}
int EigenQuasiNewton::mult(Vec *x, Vec *y) {  // :189-196
  PO_TRY(k_fill(ctx, y->d, n, 0.0));
  return multAdd(1.0, x, y);
}

// ================================================================================================
// subproblems
// ================================================================================================
TrustRegionSubproblem::TrustRegionSubproblem(Problem *p)
    : Problem(p->ctx, p->nlocal, p->ncon, p->ninequality), prob(p), m(p->ncon), xk(nullptr), lk(nullptr),
      uk(nullptr), lb(nullptr), ub(nullptr), gk(nullptr), gt(nullptr), t(nullptr), xtemp(nullptr), fk(0.0),
      ft(0.0), ck(p->ncon, 0.0), ct(p->ncon, 0.0), qn_update_type(0) {
  offset = p->offset;
  nglobal = p->nglobal;
  nwcon = p->nwcon;
  nwinequality = p->nwinequality;
}
TrustRegionSubproblem::~TrustRegionSubproblem() {
  Vec *all[] = {xk, lk, uk, lb, ub, gk, gt, t, xtemp};
  for (Vec *v : all) vec_decref(v);
  for (Vec *v : Ak) vec_decref(v);
  for (Vec *v : At) vec_decref(v);
}
int TrustRegionSubproblem::allocate() {
  Vec **all[] = {&xk, &lk, &uk, &lb, &ub, &gk, &gt, &t, &xtemp};
  for (Vec **v : all) {
    *v = vec_new(ctx, nlocal);
    if (!*v) return PO_ERR_HIP;
  }
  for (int i = 0; i < m; i++) {
    Vec *a = vec_new(ctx, nlocal), *b = vec_new(ctx, nlocal);
    if (!a || !b) return PO_ERR_HIP;
    Ak.push_back(a);
    At.push_back(b);
  }
  // defaults before initialize (:62-67)
  PO_TRY(k_fill(ctx, lk->d, nlocal, 0.0));
  PO_TRY(k_fill(ctx, uk->d, nlocal, 1.0));
  PO_TRY(k_fill(ctx, lb->d, nlocal, 0.0));
  PO_TRY(k_fill(ctx, ub->d, nlocal, 1.0));
  PO_TRY(k_fill(ctx, xk->d, nlocal, 0.5));
  return PO_OK;
}
int TrustRegionSubproblem::initModelAndBounds(double tr_size) {  // :141-151
  zts_valid = false;
  if (prob->getVarsAndBounds(xk, lb, ub) != 0) return PO_ERR_USER;
  PO_TRY(setTrustRegionBounds(tr_size));
  if (prob->evalObjCon(xk, &fk, ck.data()) != 0) return PO_ERR_USER;
  if (prob->evalObjConGradient(xk, gk, Ak.data()) != 0) return PO_ERR_USER;
  return PO_OK;
}
int TrustRegionSubproblem::setTrustRegionBounds(double tr_size) {
  return k_tr_bounds(ctx, xk->d, lb->d, ub->d, tr_size, nlocal, lk->d, uk->d);
}
int TrustRegionSubproblem::getVarsAndBounds(Vec *x, Vec *l, Vec *u) {  // :278-285
  const double half[2] = {0.5, 0.5};
  const double *vv[2] = {lk->d, uk->d};
  PO_TRY(k_panel_axpy(ctx, x->d, 0.0, nullptr, 0.0, half, vv, 2, nlocal));
  PO_TRY(k_copy(ctx, l->d, lk->d, nlocal));
  PO_TRY(k_copy(ctx, u->d, uk->d, nlocal));
  return 0;
}
int TrustRegionSubproblem::evalSparseCon(Vec *step, Vec *out) {  // :345-349
  if (prob->evalSparseCon(xk, out) != 0) return 1;
  return prob->addSparseJacobian(1.0, xk, step, out);
}
int TrustRegionSubproblem::addSparseJacobian(double alpha, Vec *, Vec *px, Vec *out) {
  return prob->addSparseJacobian(alpha, xk, px, out);
}
int TrustRegionSubproblem::addSparseJacobianTranspose(double alpha, Vec *, Vec *pzw, Vec *out) {
  return prob->addSparseJacobianTranspose(alpha, xk, pzw, out);
}
int TrustRegionSubproblem::addSparseInnerProduct(double alpha, Vec *, Vec *cvec, Vec *A) {
  return prob->addSparseInnerProduct(alpha, xk, cvec, A);
}
int TrustRegionSubproblem::sparseJacobianPanel(Vec *, Vec *d, const double *const *P, int nv,
                                               double *const *U, Vec *work) {
  return prob->sparseJacobianPanel(xk, d, P, nv, U, work);
}
void TrustRegionSubproblem::rejectTrialStep() {  // :226-231
  ft = 0.0;
  for (int i = 0; i < m; i++) ct[i] = 0.0;
}
int TrustRegionSubproblem::evalLinearModel(Vec *step, double *f, double *cons) {
  std::vector<const double *> P;
  P.push_back(gk->d);
  for (Vec *a : Ak) P.push_back(a->d);
  lin_dots.assign(m + 1, 0.0);
  BatchScope batch(ctx);
  PO_TRY(k_mdot(ctx, step->d, P.data(), m + 1, nlocal, lin_dots.data()));
  return batch.end_then([this, f, cons] {
    *f = fk + lin_dots[0];
    for (int i = 0; i < m; i++) cons[i] = ck[i] + lin_dots[1 + i];
  });
}
int TrustRegionSubproblem::evalTrialPoint(Vec *step, double *fobj, double *cons) {  // :178-188
  const double one[1] = {1.0};
  const double *vv[1] = {step->d};
  PO_TRY(k_panel_axpy(ctx, xtemp->d, 1.0, xk->d, 0.0, one, vv, 1, nlocal));
  int fail = prob->evalObjCon(xtemp, &ft, ct.data());
  fail = fail || prob->evalObjConGradient(xtemp, gt, At.data());
  *fobj = ft;
  for (int i = 0; i < m; i++) cons[i] = ct[i];
  return fail ? PO_ERR_USER : PO_OK;
}
// t = [gt - At^T z - Aw(xtemp)^T zw] - [gk - Ak^T z - Aw(xk)^T zw]
int TrustRegionSubproblem::lagrangianGradientDifference(const double *z, Vec *zw) {
  std::vector<const double *> P;
  std::vector<double> cf;
  for (int i = 0; i < m; i++) {
    P.push_back(At[i]->d);
    cf.push_back(-z[i]);
  }
  P.push_back(gk->d);
  cf.push_back(-1.0);
  for (int i = 0; i < m; i++) {
    P.push_back(Ak[i]->d);
    cf.push_back(z[i]);
  }
  PO_TRY(k_panel_axpy(ctx, t->d, 1.0, gt->d, 0.0, cf.data(), P.data(), (int)P.size(), nlocal));
  if (nwcon > 0 && zw) {
    if (prob->addSparseJacobianTranspose(-1.0, xtemp, zw, t) != 0) return PO_ERR_USER;
    if (prob->addSparseJacobianTranspose(1.0, xk, zw, t) != 0) return PO_ERR_USER;
  }
  return PO_OK;
}
void TrustRegionSubproblem::acceptModel() {
  zts_valid = false;
  fk = ft;
  std::swap(gk->d, gt->d);
  for (int i = 0; i < m; i++) {
    ck[i] = ct[i];
    std::swap(Ak[i]->d, At[i]->d);
  }
}

// ---- user-written subproblem behind a callback table ----------------------------------------------------------
static inline po_vec PV(Vec *v) { return static_cast<po_vec>(v); }
void CallbackSubproblem::dropModel() {
  Vec **all[] = {&xk, &gk, &lb, &ub};
  for (Vec **v : all) {
    vec_decref(*v);
    *v = nullptr;
  }
  for (Vec *v : Ak) vec_decref(v);
  Ak.clear();
}
CallbackSubproblem::~CallbackSubproblem() { dropModel(); }
int CallbackSubproblem::allocateModel() {
  // only the driver's scratch that no callback provides; the model vectors are borrowed in syncLinearModel
  t = vec_new(ctx, nlocal);
  xtemp = vec_new(ctx, nlocal);
  return (t && xtemp) ? PO_OK : PO_ERR_HIP;
}
int CallbackSubproblem::syncLinearModel() {
  po_vec hx = nullptr, hg = nullptr, hl = nullptr, hu = nullptr;
  const double *c = nullptr;
  const po_vec *A = nullptr;
  double f = 0.0;
  if (!cb.get_linear_model || cb.get_linear_model(cb.user, &hx, &f, &hg, &c, &A, &hl, &hu) != 0) {
    set_error("trust-region subproblem: getLinearModel failed");
    return PO_ERR_USER;
  }
  if (!hx || !hg || !hl || !hu || (m > 0 && (!c || !A))) {
    set_error("trust-region subproblem: getLinearModel returned a NULL part of the model");
    return PO_ERR_USER;
  }
  dropModel();
  // The user's vectors may carry LIVE host mirrors (filled through getArray pointers, e.g. by the original problem's
  // host-array callbacks): the kernels below read the device arrays directly, so what the host wrote is uploaded here,
  // as po_vec_release_array(v, 1) would (src/ParOptVec.cpp:212-217: the pointer is the data).
  int up_rc = PO_OK;
  auto take = [&](po_vec h) -> Vec * {
    Vec *v = h;
    v->ref++;
    if (v->h_live && v->h && v->n > 0 &&
        hipMemcpyAsync(v->d, v->h, sizeof(double) * (size_t)v->n, hipMemcpyHostToDevice, ctx->stream) != hipSuccess)
      up_rc = PO_ERR_HIP;
    return v;
  };
  xk = take(hx);
  gk = take(hg);
  lb = take(hl);
  ub = take(hu);
  fk = f;
  for (int i = 0; i < m; i++) {
    if (!A[i] || A[i]->n != nlocal) {
      set_error("trust-region subproblem: getLinearModel: constraint gradient %d is missing or has the wrong size", i);
      return PO_ERR_USER;
    }
    ck[i] = c[i];
    Ak.push_back(take(A[i]));
  }
  zts_valid = false;
  if (up_rc != PO_OK) {
    set_error("trust-region subproblem: uploading a live host mirror of the linear model failed");
    return up_rc;
  }
  PO_HIP(hipStreamSynchronize(ctx->stream));  // the pinned mirrors may be rewritten by the user at once
  return PO_OK;
}
CompactQuasiNewton *CallbackSubproblem::getQuasiNewton() {
  po_qn q = nullptr;
  if (cb.get_quasi_newton && cb.get_quasi_newton(cb.user, &q) == 0 && q) return q->qn;
  return nullptr;
}
int CallbackSubproblem::initModelAndBounds(double tr_size) {
  if (!cb.init_model_and_bounds || cb.init_model_and_bounds(cb.user, tr_size) != 0) return PO_ERR_USER;
  return syncLinearModel();
}
int CallbackSubproblem::setTrustRegionBounds(double tr_size) {
  return (cb.set_trust_region_bounds && cb.set_trust_region_bounds(cb.user, tr_size) == 0) ? PO_OK : PO_ERR_USER;
}
int CallbackSubproblem::evalTrialStepAndUpdate(int update_flag, Vec *step, const double *z, Vec *zw, double *fobj,
                                               double *cons) {
  if (!cb.eval_trial_step_and_update) return PO_ERR_USER;
  return cb.eval_trial_step_and_update(cb.user, update_flag, PV(step), z, PV(zw), fobj, cons) == 0 ? PO_OK : PO_ERR_USER;
}
int CallbackSubproblem::acceptTrialStep(Vec *step, const double *z, Vec *zw) {
  if (!cb.accept_trial_step || cb.accept_trial_step(cb.user, PV(step), z, PV(zw)) != 0) return PO_ERR_USER;
  return syncLinearModel();
}
void CallbackSubproblem::rejectTrialStep() {
  if (cb.reject_trial_step) cb.reject_trial_step(cb.user);
}
int CallbackSubproblem::getQuasiNewtonUpdateType() const {
  return cb.get_quasi_newton_update_type ? cb.get_quasi_newton_update_type(cb.user) : 0;
}
int CallbackSubproblem::getVarsAndBounds(Vec *x, Vec *l, Vec *u) {
  return cb.get_vars_and_bounds ? cb.get_vars_and_bounds(cb.user, PV(x), PV(l), PV(u)) : 1;
}
int CallbackSubproblem::evalObjCon(Vec *step, double *fobj, double *cons) {
  return cb.eval_obj_con ? cb.eval_obj_con(cb.user, PV(step), fobj, cons) : 1;
}
int CallbackSubproblem::evalObjConGradient(Vec *step, Vec *g, Vec **Ac) {
  if (!cb.eval_obj_con_gradient) return 1;
  std::vector<po_vec> h(m > 0 ? m : 1, nullptr);
  for (int i = 0; Ac && i < m; i++) h[i] = PV(Ac[i]);
  return cb.eval_obj_con_gradient(cb.user, PV(step), PV(g), (Ac || m == 0) ? h.data() : nullptr);
}

// ---- quadratic --------------------------------------------------------------------------------
int QuadraticSubproblem::evalTrialStepAndUpdate(int update_flag, Vec *step, const double *z, Vec *zw,
                                                double *fobj, double *cons) {  // :175-212
  zts_valid = false;
  PO_TRY(evalTrialPoint(step, fobj, cons));
  if (qn && update_flag) {
    PO_TRY(lagrangianGradientDifference(z, zw));
    if (prob->computeQuasiNewtonUpdateCorrection(xtemp, z, step, t) != 0) return PO_ERR_USER;
    PO_TRY(qn->update(step, t, &qn_update_type));
  }
  return PO_OK;
}
int QuadraticSubproblem::acceptTrialStep(Vec *step, const double *, Vec *) {  // :214-224
  PO_TRY(k_axpy(ctx, xk->d, 1.0, step->d, nlocal));
  acceptModel();
  return PO_OK;
}
int QuadraticSubproblem::evalObjCon(Vec *step, double *fobj, double *cons) {  // :290-323
  if (!step) {
    *fobj = fk;
    for (int i = 0; i < m; i++) cons[i] = ck[i];
    return 0;
  }
  std::vector<const double *> P;
  P.push_back(gk->d);
  for (Vec *a : Ak) P.push_back(a->d);
  int k = 0;
  if (qn) {
    std::vector<const double *> zp = qn->zPointers();
    k = (int)zp.size();
    P.insert(P.end(), zp.begin(), zp.end());
    P.push_back(step->d);
  }
  eo_dots.assign(P.size(), 0.0);
  zts_valid = false;
  BatchScope batch(ctx);
  if (k_mdot(ctx, step->d, P.data(), (int)P.size(), nlocal, eo_dots.data()) != PO_OK) return 1;
  const double *sd = step->d;
  const int rc = batch.end_then([this, fobj, cons, k, sd] {
    const std::vector<double> &dots = eo_dots;
    double f = fk + dots[0];
    if (qn) {
      std::vector<double> rz(dots.begin() + 1 + m, dots.begin() + 1 + m + k), cf = rz;
      if (k > 0) qn->applyCompactInverse(cf.data());
      double sBs = qn->diag() * dots[1 + m + k];
      for (int i = 0; i < k; i++) sBs -= rz[i] * cf[i];
      f += 0.5 * sBs;
      zts_cache = rz;  // Z^T step for the gradient evaluation at this very step
      zts_ptr = sd;
      zts_valid = true;
    }
    *fobj = f;
    for (int i = 0; i < m; i++) cons[i] = ck[i] + dots[1 + i];
  });
  return rc == PO_OK ? 0 : 1;
}
int QuadraticSubproblem::evalObjConGradient(Vec *step, Vec *g, Vec **Ac) {  // :328-343
  // Ac == nullptr: the constraint model is linear (linear_constraints is set), the solver kept the Jacobian of the
  // first evaluation of this solve
  std::vector<double *> dst;
  std::vector<const double *> src;
  for (int i = 0; Ac && i < m; i++) {
    dst.push_back(Ac[i]->d);
    src.push_back(Ak[i]->d);
  }
  if (Ac && m > 0 && k_panel_lincomb(ctx, dst.data(), 1.0, src.data(), 0.0, nullptr, m, nlocal) != PO_OK) return 1;
  if (!qn) return k_copy(ctx, g->d, gk->d, nlocal) == PO_OK ? 0 : 1;
  std::vector<const double *> zp = qn->zPointers();
  const int k = (int)zp.size();
  std::vector<double> cf(k + 1, 0.0);
  if (k > 0) {
    if (zts_valid && zts_ptr == step->d && (int)zts_cache.size() == k) {
      for (int i = 0; i < k; i++) cf[1 + i] = zts_cache[i];  // the products evalObjCon(step) just took
    } else if (k_mdot(ctx, step->d, zp.data(), k, nlocal, cf.data() + 1) != PO_OK) {
      return 1;
    }
    qn->applyCompactInverse(cf.data() + 1);
    for (int i = 0; i < k; i++) cf[1 + i] = -cf[1 + i];
  }
  cf[0] = qn->diag();
  std::vector<const double *> P;
  P.push_back(step->d);
  P.insert(P.end(), zp.begin(), zp.end());
  return k_panel_axpy(ctx, g->d, 1.0, gk->d, 0.0, cf.data(), P.data(), k + 1, nlocal) == PO_OK ? 0 : 1;
}

// ---- compact eigenvalue -------------------------------------------------------------------------
int EigenSubproblem::initModelAndBounds(double tr_size) {  // :412-439
  PO_TRY(TrustRegionSubproblem::initModelAndBounds(tr_size));
  if (update_model) {
    CompactEigenApprox *e = approx->eigh;
    e->c0 = ck[approx->index];
    PO_TRY(k_copy(ctx, e->g0->d, Ak[approx->index]->d, nlocal));
    if (update_model(update_user, xk, e) != 0) return PO_ERR_USER;
  }
  return PO_OK;
}
int EigenSubproblem::evalTrialStepAndUpdate(int, Vec *step, const double *, Vec *, double *fobj,
                                            double *cons) {  // :460-476
  zts_valid = false;
  return evalTrialPoint(step, fobj, cons);
}
int EigenSubproblem::acceptTrialStep(Vec *step, const double *z, Vec *zw) {  // :478-529
  zts_valid = false;
  const double one[1] = {1.0};
  const double *vv[1] = {step->d};
  PO_TRY(k_panel_axpy(ctx, xtemp->d, 1.0, xk->d, 0.0, one, vv, 1, nlocal));
  if (update_model) {
    CompactEigenApprox *e = approx->eigh;
    e->c0 = ct[approx->index];
    PO_TRY(k_copy(ctx, e->g0->d, At[approx->index]->d, nlocal));
    if (update_model(update_user, xtemp, e) != 0) return PO_ERR_USER;
  }
  CompactQuasiNewton *q = approx->qn;
  if (q && z) {  // the filter strategy passes no multipliers (:1932): no curvature pair then
    PO_TRY(lagrangianGradientDifference(z, zw));
    if (prob->computeQuasiNewtonUpdateCorrection(xtemp, z, step, t) != 0) return PO_ERR_USER;
    int rc = 0;
    PO_TRY(q->update(step, t, &rc));
  }
  std::swap(xk->d, xtemp->d);
  acceptModel();
  return PO_OK;
}
// dots of `step` with [gk | Ak | Z_qn (if used) | H | g0 | step]
int EigenSubproblem::modelDots(Vec *step, std::vector<double> &dots, int *kq) {
  std::vector<const double *> P;
  P.push_back(gk->d);
  for (Vec *a : Ak) P.push_back(a->d);
  std::vector<const double *> zp = approx->zPointers();  // [Z_qn | H]
  *kq = (int)zp.size() - approx->eigh->N;
  P.insert(P.end(), zp.begin(), zp.end());
  P.push_back(approx->eigh->g0->d);
  P.push_back(step->d);
  dots.assign(P.size(), 0.0);
  return k_mdot(ctx, step->d, P.data(), (int)P.size(), nlocal, dots.data());
}
int EigenSubproblem::evalObjCon(Vec *step, double *fobj, double *cons) {  // :585-621
  CompactEigenApprox *e = approx->eigh;
  const int idx = approx->index, N = e->N;
  if (!step) {
    *fobj = fk;
    for (int i = 0; i < m; i++) cons[i] = ck[i];
    cons[idx] = e->c0;
    return 0;
  }
  int kq = 0;
  zts_valid = false;
  BatchScope batch(ctx);
  if (modelDots(step, eo_dots, &kq) != PO_OK) return 1;
  const double *sd = step->d;
  const int rc = batch.end_then([this, fobj, cons, kq, sd, e, idx, N] {
    const std::vector<double> &dots = eo_dots;
    const int k = kq + N;
    const double *rz = dots.data() + 1 + m;
    std::vector<double> cf(rz, rz + k);
    approx->applyCompactInverse(cf.data());
    double sBs = approx->diag() * dots[1 + m + k + 1];
    for (int i = 0; i < k; i++) sBs -= rz[i] * cf[i];
    *fobj = fk + dots[0] + 0.5 * sBs;
    for (int i = 0; i < m; i++) cons[i] = ck[i] + dots[1 + i];
    const double *h = rz + kq;  // H^T s
    double c = e->c0 + dots[1 + m + k];
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) c += 0.5 * e->M[(size_t)i * N + j] * h[i] * h[j];
    cons[idx] = c;
    zts_cache.assign(rz, rz + k);  // [Z_qn | H]^T step for the gradient evaluation at this very step
    zts_ptr = sd;
    zts_valid = true;
  });
  return rc == PO_OK ? 0 : 1;
}
int EigenSubproblem::evalObjConGradient(Vec *step, Vec *g, Vec **Ac) {  // :626-643
  CompactEigenApprox *e = approx->eigh;
  const int idx = approx->index, N = e->N;
  std::vector<const double *> zp = approx->zPointers();
  const int k = (int)zp.size(), kq = k - N;
  std::vector<double> rz(k > 0 ? k : 1, 0.0);
  if (k > 0 && zts_valid && zts_ptr == step->d && (int)zts_cache.size() == k) {
    for (int i = 0; i < k; i++) rz[i] = zts_cache[i];  // the products evalObjCon(step) just took
  } else if (k > 0 && k_mdot(ctx, step->d, zp.data(), k, nlocal, rz.data()) != PO_OK) {
    return 1;
  }
  // constraint gradients: copies, except the modelled one: g0 + H (M H^T s)
  std::vector<double *> dst;
  std::vector<const double *> src;
  for (int i = 0; Ac && i < m; i++) {  // (Ac == nullptr: objective gradient only, asked for by a linearised wrapper)
    if (i == idx || !Ac[i]) continue;  // (Ac[i] == nullptr: the solver kept this constant column, constantJacobianMask)
    dst.push_back(Ac[i]->d);
    src.push_back(Ak[i]->d);
  }
  if (!dst.empty() &&
      k_panel_lincomb(ctx, dst.data(), 1.0, src.data(), 0.0, nullptr, (int)dst.size(), nlocal) != PO_OK)
    return 1;
  // g = gk + B s
  std::vector<double> cf(k + 1, 0.0);
  for (int i = 0; i < k; i++) cf[1 + i] = rz[i];
  approx->applyCompactInverse(cf.data() + 1);
  for (int i = 0; i < k; i++) cf[1 + i] = -cf[1 + i];
  cf[0] = approx->diag();
  std::vector<const double *> P;
  P.push_back(step->d);
  P.insert(P.end(), zp.begin(), zp.end());
  if (Ac && Ac[idx] && k + 1 <= kMaxPanel) {
    // ... and the modelled constraint's gradient g0 + H (M H^T s) in the same pass over [step | Z_qn | H]
    std::vector<double> ch(k + 1, 0.0);
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) ch[1 + kq + i] += e->M[(size_t)i * N + j] * rz[kq + j];
    return k_panel_axpy2(ctx, g->d, 1.0, gk->d, cf.data(), Ac[idx]->d, 1.0, e->g0->d, ch.data(), P.data(), k + 1,
                         nlocal) == PO_OK ? 0 : 1;
  }
  if (Ac && Ac[idx]) {
    std::vector<double> mh(N, 0.0);
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) mh[i] += e->M[(size_t)i * N + j] * rz[kq + j];
    std::vector<const double *> hp = e->hPointers();
    if (k_panel_axpy(ctx, Ac[idx]->d, 1.0, e->g0->d, 0.0, mh.data(), hp.data(), N, nlocal) != PO_OK) return 1;
  }
  return k_panel_axpy(ctx, g->d, 1.0, gk->d, 0.0, cf.data(), P.data(), k + 1, nlocal) == PO_OK ? 0 : 1;
}

// ---- infeasibility (steering) problem -----------------------------------------------------------
InfeasSubproblem::InfeasSubproblem(TrustRegionSubproblem *sub_, int objective_, int constraint_)
    : Problem(sub_->ctx, sub_->nlocal, sub_->ncon, sub_->ninequality), sub(sub_), objective(objective_),
      constraint(constraint_), obj_scale(1.0) {
  offset = sub_->offset;
  nglobal = sub_->nglobal;
  nwcon = sub_->nwcon;
  nwinequality = sub_->nwinequality;
  // a linear constraint model has a constant Jacobian within a solve: the interior point keeps it (Ac == nullptr
  // in later gradient calls) and carries A^T z by recurrence (po_problem_set_linear_constraints)
  linear_constraints = (constraint_ == LINEAR_CONSTRAINT || sub_->linear_constraints) ? 1 : 0;
}
int InfeasSubproblem::evalObjCon(Vec *step, double *fobj, double *cons) {  // :541-580
  const int m = sub->m;
  if (!step) {
    *fobj = sub->fk * obj_scale;
    for (int i = 0; i < m; i++) cons[i] = sub->ck[i];
    return 0;
  }
  eo_cs.assign(m > 0 ? m : 1, 0.0);
  eo_cl.assign(m > 0 ? m : 1, 0.0);
  eo_fs = eo_fl = 0.0;
  const bool need_sub = objective == SUBPROBLEM_OBJECTIVE || constraint == SUBPROBLEM_CONSTRAINT;
  const bool need_lin = objective == LINEAR_OBJECTIVE || constraint == LINEAR_CONSTRAINT;
  // both model evaluations (and the caller's reductions, if it has a batch open) share one collective + sync
  BatchScope batch(ctx, sub->reductionsBatchable());
  if (need_sub && sub->evalObjCon(step, &eo_fs, eo_cs.data()) != 0) return 1;
  if (need_lin && sub->evalLinearModel(step, &eo_fl, eo_cl.data()) != PO_OK) return 1;
  const int rc = batch.end_then([this, fobj, cons, m] {
    double f = eo_fs;
    if (objective == LINEAR_OBJECTIVE) f = eo_fl;
    if (objective == CONSTANT_OBJECTIVE) f = sub->fk;
    for (int i = 0; i < m; i++) cons[i] = constraint == LINEAR_CONSTRAINT ? eo_cl[i] : eo_cs[i];
    *fobj = f * obj_scale;
  });
  return rc == PO_OK ? 0 : 1;
}
int InfeasSubproblem::evalObjConGradient(Vec *step, Vec *g, Vec **Ac) {  // :585-612
  const int m = sub->m;
  if (objective == SUBPROBLEM_OBJECTIVE || constraint == SUBPROBLEM_CONSTRAINT) {
    if (sub->evalObjConGradient(step, g, Ac) != 0) return 1;
  }
  if (objective == LINEAR_OBJECTIVE) {
    // g = obj_scale * gk in one pass
    if (k_panel_axpy(ctx, g->d, obj_scale, sub->gk->d, 0.0, nullptr, nullptr, 0, nlocal) != PO_OK) return 1;
  } else if (objective == CONSTANT_OBJECTIVE) {
    if (k_fill(ctx, g->d, nlocal, 0.0) != PO_OK) return 1;
  } else {
    if (k_scale(ctx, g->d, nlocal, obj_scale) != PO_OK) return 1;
  }
  if (constraint == LINEAR_CONSTRAINT && m > 0 && Ac) {
    std::vector<double *> dst;
    std::vector<const double *> src;
    for (int i = 0; i < m; i++) {
      dst.push_back(Ac[i]->d);
      src.push_back(sub->Ak[i]->d);
    }
    if (k_panel_lincomb(ctx, dst.data(), 1.0, src.data(), 0.0, nullptr, m, nlocal) != PO_OK) return 1;
  }
  return 0;
}

// ================================================================================================
// the driver
// ================================================================================================
TrustRegion::TrustRegion(Problem *p)
    : prob(p), ctx(p->ctx), qn(nullptr), eigh(nullptr), eqn(nullptr), sub(nullptr), infeas(nullptr), ip(nullptr),
      m(p->ncon), nineq(p->ninequality), tr_size(0.1), iter_count(0), subproblem_iters(0),
      adaptive_subproblem_iters(0), iter_cb(nullptr), iter_cb_user(nullptr), eig_N(0), eig_index(0),
      eig_update(nullptr), eig_user(nullptr), tvec(nullptr) {
  qn_handle.qn = nullptr;
  opts.addTrustRegionDefaults();
  // the inner interior-point solves do not write an iteration table unless asked to
  opts.set("output_file", "");
  for (double &v : row) v = 0.0;
}
TrustRegion::TrustRegion(TrustRegionSubproblem *s)
    : prob(s->prob), ctx(s->ctx), qn(nullptr), eigh(nullptr), eqn(nullptr), sub(s), infeas(nullptr), ip(nullptr),
      m(s->ncon), nineq(s->ninequality), tr_size(0.1), iter_count(0), subproblem_iters(0),
      adaptive_subproblem_iters(0), iter_cb(nullptr), iter_cb_user(nullptr), own_sub(false), eig_N(0), eig_index(0),
      eig_update(nullptr), eig_user(nullptr), tvec(nullptr) {
  qn_handle.qn = nullptr;
  opts.addTrustRegionDefaults();
  opts.set("output_file", "");
  for (double &v : row) v = 0.0;
}
TrustRegion::~TrustRegion() {
  if (own_ip) delete ip;
  delete infeas;
  if (own_sub) delete sub;
  delete eqn;
  delete eigh;
  delete qn;
  vec_decref(tvec);
}
int TrustRegion::setEigenModel(int N, int index, EigenModelUpdate update, void *user) {
  if (sub) {
    set_error("the eigenvalue model must be configured before the first optimize");
    return PO_ERR_ARG;
  }
  if (N < 1 || N > 64 || index < 0 || index >= m) {
    set_error("bad eigenvalue model (N=%d, index=%d, ncon=%d)", N, index, m);
    return PO_ERR_ARG;
  }
  eig_N = N;
  eig_index = index;
  eig_update = update;
  eig_user = user;
  return PO_OK;
}

// the set-up of ParOptOptimizer::optimize for algorithm = "tr" (src/ParOptOptimizer.cpp:108-183)
int TrustRegion::initState() {
  if (state_ready) return PO_OK;
  penalty_gamma.assign(m, options().real("penalty_gamma"));
  tr_size = options().real("tr_init_size");
  state_ready = true;
  return PO_OK;
}
void TrustRegion::setPenaltyGamma(double gamma) {
  initState();
  if (gamma >= 0.0) penalty_gamma.assign(m, gamma);
}
void TrustRegion::setPenaltyGammaArray(const double *g) {
  initState();
  for (int i = 0; i < m; i++)
    if (g[i] >= 0.0) penalty_gamma[i] = g[i];
}
int TrustRegion::initialize() {
  PO_TRY(build());
  PO_TRY(sub->initModelAndBounds(tr_size));
  iter_count = 0;
  return PO_OK;
}

int TrustRegion::build() {
  if (ip) return initState();
  const int64_t n = prob->nlocal;
  if (!own_sub) {
    // the caller's subproblem (allocated by its creator); an interior-point solver of our own only if optimize() is
    // called without one
    if (!tvec) tvec = vec_new(ctx, n);
    if (!tvec) return PO_ERR_HIP;
    ip = new InteriorPoint(sub);
    ip->options = opts;
    PO_TRY(ip->allocate());
    own_ip = true;
    qn_handle.qn = sub->getQuasiNewton();
    return initState();
  }
  const std::string qt = opts.str("qn_type");
  const int msub = opts.integer("qn_subspace_size");
  if (qt == "bfgs") {
    LBFGS *b = new LBFGS(ctx, n, msub);
    b->setBFGSUpdateType(std::string(opts.str("qn_update_type")) == "damped_update"
                             ? PO_BFGS_DAMPED_UPDATE
                             : PO_BFGS_SKIP_NEGATIVE_CURVATURE);
    qn = b;
  } else if (qt == "sr1") {
    qn = new LSR1(ctx, n, msub);
  }
  if (qn) {
    const std::string dt = opts.str("qn_diag_type");
    qn->setInitDiagonalType(dt == "yts_over_sts" ? PO_QN_YTS_OVER_STS : PO_QN_YTY_OVER_YTS);
  }
  if (eig_N > 0) {
    eigh = new CompactEigenApprox(ctx, n, eig_N);
    PO_TRY(eigh->allocate());
    eqn = new EigenQuasiNewton(qn, eigh, eig_index);
    EigenSubproblem *es = new EigenSubproblem(prob, eqn);
    es->update_model = eig_update;
    es->update_user = eig_user;
    sub = es;
  } else {
    sub = new QuadraticSubproblem(prob, qn);
  }
  PO_TRY(sub->allocate());
  tvec = vec_new(ctx, n);
  if (!tvec) return PO_ERR_HIP;
  ip = new InteriorPoint(sub);
  ip->options = opts;
  PO_TRY(ip->allocate());
  qn_handle.qn = sub->getQuasiNewton();
  return initState();
}

double TrustRegion::infeasOf(const double *c, const double *weights) const {
  double tot = 0.0;
  for (int i = 0; i < m; i++) {
    const double v = i < nineq ? std::max(0.0, -c[i]) : fabs(c[i]);
    tot += weights ? weights[i] * v : v;
  }
  return tot;
}

int TrustRegion::computeKKTError(const double *z, Vec *zw, double *l1, double *linfty) {  // :2391-2472
  const int64_t n = prob->nlocal;
  std::vector<const double *> P;
  std::vector<double> cf;
  for (int i = 0; i < m; i++) {
    P.push_back(sub->Ak[i]->d);
    cf.push_back(-z[i]);
  }
  PO_TRY(k_panel_axpy(ctx, tvec->d, 1.0, sub->gk->d, 0.0, cf.data(), P.data(), m, n));
  if (sub->nwcon > 0 && zw) {
    if (sub->addSparseJacobianTranspose(-1.0, sub->xk, zw, tvec) != 0) return PO_ERR_USER;
  }
  double out[2];
  PO_TRY(k_kkt_error(ctx, sub->xk->d, sub->lb->d, sub->ub->d, tvec->d, options().real("tr_bound_relax"), n, out));
  double zmax = 0.0;
  if (sub->nwcon > 0 && zw) PO_TRY(k_reduce1(ctx, RED_AMAX, zw->d, nullptr, zw->n, &zmax));
  for (int i = 0; i < m; i++) zmax = std::max(zmax, fabs(z[i]));
  zmax = std::max(1.0, zmax);
  double g1 = 0.0, ginf = 0.0;
  PO_TRY(k_reduce1(ctx, RED_ASUM, sub->gk->d, nullptr, n, &g1));
  PO_TRY(k_reduce1(ctx, RED_AMAX, sub->gk->d, nullptr, n, &ginf));
  *l1 = out[0] / std::max(g1, zmax);
  *linfty = out[1] / std::max(ginf, zmax);
  return PO_OK;
}

int TrustRegion::acceptableByPair(double f_new, double h_new, double f_old, double h_old) {  // :896-922
  const double gamma = options().real("filter_gamma");
  double _f_old = f_old, _h_old = h_old;
  if (options().integer("filter_sufficient_reduction")) {
    _h_old = (1.0 - gamma) * h_old;
    _f_old = f_old - gamma * h_new;
  }
  return (h_new < _h_old || f_new < _f_old) ? 1 : 0;
}
int TrustRegion::acceptableByFilter(double f, double h) {  // :931-939
  for (const FilterElement &e : filter)
    if (!acceptableByPair(f, h, e.f, e.h)) return 0;
  return 1;
}
void TrustRegion::addToFilter(double f, double h) {  // :947-966 (dominated pairs leave)
  std::vector<FilterElement> keep;
  for (const FilterElement &e : filter)
    if (!(f <= e.f && h <= e.h)) keep.push_back(e);
  keep.push_back(FilterElement{f, h});
  filter.swap(keep);
}

void TrustRegion::appendRow(const double vals[12], const std::string &info, double seconds) {
  for (int i = 0; i < 12; i++) row[i] = vals[i];
  row_info = info;
  if (ctx->rank != 0) return;
  char line[512];
  if (iter_count % 10 == 0) {
    snprintf(line, sizeof(line), "\n%5s %12s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %9s %-12s\n", "iter",
             "fobj", "infeas", "l1", "linfty", "|x - xk|", "tr", "rho", "mod red.", "avg z", "max z", "avg pen.",
             "max pen.", "time(s)", "info");
    history += line;
  }
  snprintf(line, sizeof(line),
           "%5d %12.5e %9.2e %9.2e %9.2e %9.2e %9.2e %9.2e %9.2e %9.2e %9.2e %9.2e %9.2e %9.2e %-12s\n", iter_count,
           vals[0], vals[1], vals[2], vals[3], vals[4], vals[5], vals[6], vals[7], vals[8], vals[9], vals[10],
           vals[11], seconds, info.c_str());
  history += line;
}

int TrustRegion::minimizeInfeas(std::vector<double> *best_out) {  // :1105-1228
  Options &o = ip->options;
  const std::string start_option = o.str("starting_point_strategy");
  const std::string barrier_option = o.str("barrier_strategy");
  const std::string tr_barrier = o.str("tr_steering_barrier_strategy");
  const std::string tr_start = o.str("tr_steering_starting_point_strategy");
  PO_TRY(ip->resetProblemInstance(infeas));
  if (tr_barrier != "default") PO_TRY(o.set("barrier_strategy", tr_barrier.c_str()));
  if (tr_start != "default") PO_TRY(o.set("starting_point_strategy", tr_start.c_str()));
  const int is_seq = o.integer("sequential_linear_method");
  // (:1150-1156: whatever compact approximation the subproblem carries, the caller's included)
  EigenQuasiNewton *eq = eqn ? eqn : dynamic_cast<EigenQuasiNewton *>(sub->getQuasiNewton());
  if (infeas->objective == InfeasSubproblem::LINEAR_OBJECTIVE ||
      infeas->objective == InfeasSubproblem::CONSTANT_OBJECTIVE) {
    if (eq) eq->use_qn_objective = 0;
    if (infeas->constraint == InfeasSubproblem::LINEAR_CONSTRAINT) PO_TRY(o.set("sequential_linear_method", 1));
  }
  double gamma = 1e6;
  const double gmax = o.real("tr_penalty_gamma_max");
  if (1e2 * gmax > gamma) gamma = 1e2 * gmax;
  infeas->obj_scale = 1.0 / gamma;
  ip->setPenaltyGamma(1.0);
  PO_TRY(ip->resetDesignAndBounds());
  int rc = ip->optimize(nullptr);
  if (rc != 0 && rc != 1) return rc;
  captureSolveLine(0);
  Vec *step = nullptr;
  ip->getOptimizedPoint(&step, nullptr, nullptr, nullptr);
  if (std::string(o.str("tr_accept_step_strategy")) == "penalty_method" && o.integer("tr_adaptive_gamma_update")) {
    ip->getIterationCounters(&adaptive_subproblem_iters, nullptr, nullptr);
  }
  if (best_out) {
    std::vector<double> &best = *best_out;
    double dummy = 0.0;
    best.assign(m > 0 ? m : 1, 0.0);
    if (sub->evalObjCon(step, &dummy, best.data()) != 0) return PO_ERR_USER;
    for (int j = 0; j < m; j++) best[j] = j < nineq ? std::max(0.0, -best[j]) : fabs(best[j]);
  }
  ip->setPenaltyGammaArray(penalty_gamma.data());
  PO_TRY(ip->resetProblemInstance(sub));
  if (eq) eq->use_qn_objective = 1;
  PO_TRY(o.set("starting_point_strategy", start_option.c_str()));
  PO_TRY(o.set("barrier_strategy", barrier_option.c_str()));
  PO_TRY(o.set("sequential_linear_method", is_seq));
  return PO_OK;
}

int TrustRegion::sl1qpUpdate(Vec *step, const double *z, Vec *zw, double *infeas_, double *l1,
                             double *linfty) {  // :1231-1443
  const double t0 = now_seconds();
  Options &o = ip->options;
  const double tr_eta = o.real("tr_eta"), tr_min = o.real("tr_min_size"), tr_max = o.real("tr_max_size");
  const double fprec = o.real("function_precision");
  const bool adaptive = o.integer("tr_adaptive_gamma_update");
  double fk = 0.0, ft = 0.0;
  std::vector<double> ck(m > 0 ? m : 1), ct(m > 0 ? m : 1);
  if (sub->evalObjCon(nullptr, &fk, ck.data()) != 0) return PO_ERR_USER;
  const double infeas_k = infeasOf(ck.data(), penalty_gamma.data());
  if (sub->evalObjCon(step, &ft, ct.data()) != 0) return PO_ERR_USER;
  const double obj_reduc = fk - ft;
  const double infeas_model = infeasOf(ct.data(), penalty_gamma.data());
  PO_TRY(sub->evalTrialStepAndUpdate(1, step, z, zw, &ft, ct.data()));
  const double infeas_t = infeasOf(ct.data(), penalty_gamma.data());
  const double actual_reduc = (fk - ft + (infeas_k - infeas_t));
  const double model_reduc = obj_reduc + (infeas_k - infeas_model);
  double rho = 1.0;
  if (!(fabs(model_reduc) <= fprec && fabs(actual_reduc) <= fprec)) rho = actual_reduc / model_reduc;
  *infeas_ = infeasOf(ct.data(), nullptr);
  double smax = 0.0;
  int accepted = 0;
  if (rho >= tr_eta || tr_size <= tr_min) {
    PO_TRY(k_reduce1(ctx, RED_AMAX, step->d, nullptr, prob->nlocal, &smax));
    PO_TRY(sub->acceptTrialStep(step, z, zw));
    accepted = 1;
  } else {
    sub->rejectTrialStep();
  }
  if (rho < 0.25) {
    tr_size = std::max(0.25 * tr_size, tr_min);
  } else if (rho > 0.75) {
    tr_size = std::min(1.5 * tr_size, tr_max);
  }
  PO_TRY(sub->setTrustRegionBounds(tr_size));
  PO_TRY(computeKKTError(z, zw, l1, linfty));
  double zmax = 0.0, zav = 0.0, gmax = 0.0, gav = 0.0;
  for (int i = 0; i < m; i++) {
    zav += fabs(z[i]);
    gav += penalty_gamma[i];
    zmax = std::max(zmax, fabs(z[i]));
    gmax = std::max(gmax, penalty_gamma[i]);
  }
  zav = zav / m;
  gav = gav / m;
  std::string info;
  const int ut = sub->getQuasiNewtonUpdateType();
  if (ut == 1) info += "dampH ";
  else if (ut == 2) info += "skipH ";
  char buf[64];
  if (adaptive) {
    snprintf(buf, sizeof(buf), "%d/%d ", subproblem_iters, adaptive_subproblem_iters);
  } else {
    snprintf(buf, sizeof(buf), "%d ", subproblem_iters);
  }
  info += buf;
  if (!accepted) info += "rej ";
  const double vals[12] = {fk, *infeas_, *l1, *linfty, smax, tr_size, rho, model_reduc, zav, zmax, gav, gmax};
  appendRow(vals, info, now_seconds() - t0);
  iter_count++;
  return PO_OK;
}

void TrustRegion::captureSolveLine(int which) {
  const std::string &h = ip->history;
  std::string last;
  size_t pos = 0;
  while (pos < h.size()) {
    size_t e = h.find('\n', pos);
    if (e == std::string::npos) e = h.size();
    size_t q = pos;
    while (q < e && h[q] == ' ') q++;
    if (q < e && h[q] >= '0' && h[q] <= '9') last.assign(h, pos, e - pos);
    pos = e + 1;
  }
  last_solve_line[which] = last;
}

int TrustRegion::optimize(InteriorPoint *ext) {  // optimize :2365-2384
  if (ext) {
    if (own_sub || ext->prob != sub) {
      set_error("ParOptTrustRegion::optimize: the interior-point solver was not built on this subproblem");
      return PO_ERR_ARG;
    }
    // one registry, as in the reference (the same ParOptOptions object serves both): the trust-region entries set on
    // this object move into the solver's registry; the interior-point entries are the solver's own
    Options base;
    if (ip != ext) {
      const Options &mine = ip ? ip->options : opts;
      ext->options.adoptExtras(mine, base);
      if (ip && own_ip) delete ip;
      ip = ext;
      own_ip = false;
    }
    if (!tvec) tvec = vec_new(ctx, prob->nlocal);
    if (!tvec) return PO_ERR_HIP;
    qn_handle.qn = sub->getQuasiNewton();
    PO_TRY(initState());
  }
  PO_TRY(build());
  // tr_use_soc has no effect in the reference either: the only call of isAcceptedBySoc is commented
  // out (:2002-2052), the option merely allocates a scratch vector (:702-706)
  if (std::string(ip->options.str("tr_accept_step_strategy")) == "filter_method") return filterOptimize();
  return sl1qpOptimize();
}

// filterOptimize :1690-2210, quirks included: the compatibility test after the QP looks at the LAST
// constraint only and uses |c| for inequalities (:1826-1832)
int TrustRegion::filterOptimize() {
  Options &o = ip->options;
  const int max_it = o.integer("tr_max_iterations");
  const double tr_eta = o.real("tr_eta"), tr_min = o.real("tr_min_size"), tr_max = o.real("tr_max_size");
  const double infeas_tol = o.real("tr_infeas_tol"), l1_tol = o.real("tr_l1_tol"), linf_tol = o.real("tr_linfty_tol");
  const int wfreq = o.integer("tr_write_output_frequency");
  CompactQuasiNewton *q = sub->getQuasiNewton();
  PO_TRY(ip->setQuasiNewton(q));
  PO_TRY(o.set("use_quasi_newton_update", 0));
  PO_TRY(o.set("write_output_frequency", 0));
  ip->setPenaltyGammaArray(penalty_gamma.data());
  if (!infeas) infeas = new InfeasSubproblem(sub, InfeasSubproblem::LINEAR_OBJECTIVE, InfeasSubproblem::LINEAR_CONSTRAINT);
  infeas->objective = InfeasSubproblem::LINEAR_OBJECTIVE;
  infeas->constraint = InfeasSubproblem::LINEAR_CONSTRAINT;
  history.clear();
  PO_TRY(sub->initModelAndBounds(tr_size));
  iter_count = 0;
  std::vector<double> con_trial(m > 0 ? m : 1), ck(m > 0 ? m : 1), cm(m > 0 ? m : 1);
  double fobj_init = 0.0;
  if (sub->evalObjCon(nullptr, &fobj_init, con_trial.data()) != 0) return PO_ERR_USER;
  const double infeas_init = infeasOf(con_trial.data(), nullptr);
  filter.clear();
  addToFilter(-1e20, std::max(1e4, 1.25 * infeas_init));
  int this_resto = 0, last_resto = 0;
  for (int iteration = 0; iteration < max_it; iteration++) {
    const double t0 = now_seconds();
    double fk = 0.0;
    if (sub->evalObjCon(nullptr, &fk, ck.data()) != 0) return PO_ERR_USER;
    const double hk = infeasOf(ck.data(), nullptr);
    PO_TRY(ip->resetProblemInstance(sub));
    PO_TRY(o.set("sequential_linear_method", 0));
    PO_TRY(ip->resetDesignAndBounds());
    int rc = ip->optimize(nullptr);
    if (rc != 0 && rc != 1) return rc;
    captureSolveLine(1);
    // step / z / zw alias the solver's storage, so a restoration solve below replaces them (:1797-1799)
    Vec *step = nullptr;
    const double *z = nullptr;
    ip->getOptimizedPoint(&step, &z, nullptr, nullptr);
    Vec *wv[5];
    ip->getOptimizedSparse(wv);
    Vec *zw = wv[0];
    if (o.integer("filter_has_feas_restore_phase")) {
      double dummy = 0.0, infeas_v = 0.0;
      if (sub->evalObjCon(step, &dummy, cm.data()) != 0) return PO_ERR_USER;
      for (int i = 0; i < m; i++) infeas_v = i < nineq ? std::max(0.0, fabs(-cm[i])) : fabs(cm[i]);
      if (infeas_v > infeas_tol) {
        this_resto = 1;
        addToFilter(fk, hk);
      } else {
        this_resto = 0;
        if (last_resto && q) q->reset();
      }
    }
    if (this_resto) {
      if (!last_resto && q) q->reset();
      PO_TRY(minimizeInfeas(nullptr));
    }
    double fobj_model = 0.0, fobj_trial = 0.0;
    if (sub->evalObjCon(step, &fobj_model, cm.data()) != 0) return PO_ERR_USER;
    PO_TRY(sub->evalTrialStepAndUpdate(1, step, z, zw, &fobj_trial, con_trial.data()));
    const double infeas_trial = infeasOf(con_trial.data(), nullptr);
    double smax = 0.0;
    PO_TRY(k_reduce1(ctx, RED_AMAX, step->d, nullptr, prob->nlocal, &smax));
    int init_tr = 0, inc_tr = 0, dec_tr = 0, accepted = 0;
    std::string rej;
    const double model_red = fk - fobj_model, actual_red = fk - fobj_trial;
    const double rho = actual_red / model_red;
    if (this_resto) {
      PO_TRY(sub->acceptTrialStep(step, nullptr, nullptr));
      accepted = 1;
      if (smax >= 0.99 * tr_size) inc_tr = 1;
    } else {
      const int by_filter = acceptableByFilter(fobj_trial, infeas_trial);
      const int by_pair = acceptableByPair(fobj_trial, infeas_trial, fk, hk);
      if (by_filter && by_pair) {
        if (actual_red < tr_eta * model_red && model_red > 0.0) {
          sub->rejectTrialStep();
          smax = 0.0;
          dec_tr = 1;
          rej = "rej:rho";
        } else {
          PO_TRY(sub->acceptTrialStep(step, nullptr, nullptr));
          accepted = 1;
          if (model_red <= 0.0) addToFilter(fobj_trial, infeas_trial);
          init_tr = 1;
        }
      } else if (tr_size <= tr_min) {
        PO_TRY(sub->acceptTrialStep(step, nullptr, nullptr));
        accepted = 1;
        if (smax >= 0.99 * tr_size) inc_tr = 1;
      } else {
        sub->rejectTrialStep();
        smax = 0.0;
        dec_tr = 1;
        rej = "rej:";
        if (!by_filter) rej += "F";
        if (!by_pair) rej += "xk";
      }
    }
    if (wfreq > 0 && iteration % wfreq == 0) sub->writeOutput(iteration, sub->xk);
    if (iter_cb) iter_cb(iter_cb_user, iteration);
    double l1 = 0.0, linfty = 0.0;
    PO_TRY(computeKKTError(z, zw, &l1, &linfty));
    double zmax = 0.0, zav = 0.0, gmax = 0.0, gav = 0.0;
    for (int i = 0; i < m; i++) {
      zav += fabs(z[i]);
      gav += penalty_gamma[i];
      zmax = std::max(zmax, fabs(z[i]));
      gmax = std::max(gmax, penalty_gamma[i]);
    }
    zav = zav / m;
    gav = gav / m;
    int qp_iters = 0;
    ip->getIterationCounters(&qp_iters, nullptr, nullptr);
    subproblem_iters = qp_iters;
    std::string info;
    const int ut = sub->getQuasiNewtonUpdateType();
    if (ut == 1) info += "dampH ";
    else if (ut == 2) info += "skipH ";
    char buf[64];
    snprintf(buf, sizeof(buf), "%d f%ld ", qp_iters, (long)filter.size());
    info += buf;
    if (this_resto) info += "R ";
    if (!accepted) info += (rej.empty() ? std::string("rej") : rej) + " ";
    const double vals[12] = {fobj_trial, infeas_trial, l1, linfty, smax, tr_size, rho, model_red, zav, zmax, gav, gmax};
    appendRow(vals, info, now_seconds() - t0);
    if (inc_tr) {
      tr_size = std::min(2.0 * tr_size, tr_max);
    } else if (dec_tr) {
      tr_size = std::max(0.5 * tr_size, tr_min);
    }
    if (init_tr) tr_size = tr_max;
    PO_TRY(sub->setTrustRegionBounds(tr_size));
    iter_count++;
    last_resto = this_resto;
    if (infeas_trial < infeas_tol && (l1 < l1_tol || linfty < linf_tol)) break;
  }
  flushHistory();
  return 0;
}

int TrustRegion::sl1qpOptimize() {  // :1453-1687
  Options &o = ip->options;
  const bool adaptive = o.integer("tr_adaptive_gamma_update");
  const int max_it = o.integer("tr_max_iterations");
  const double gmax = o.real("tr_penalty_gamma_max"), gmin = o.real("tr_penalty_gamma_min");
  const double infeas_tol = o.real("tr_infeas_tol"), l1_tol = o.real("tr_l1_tol"), linf_tol = o.real("tr_linfty_tol");
  const int wfreq = o.integer("tr_write_output_frequency");
  PO_TRY(ip->setQuasiNewton(sub->getQuasiNewton()));
  PO_TRY(o.set("use_quasi_newton_update", 0));
  PO_TRY(o.set("write_output_frequency", 0));
  ip->setPenaltyGammaArray(penalty_gamma.data());
  if (adaptive && !infeas) {
    const std::string ob = o.str("tr_adaptive_objective"), cn = o.str("tr_adaptive_constraint");
    const int of = ob == "constant_objective" ? InfeasSubproblem::CONSTANT_OBJECTIVE
                   : ob == "subproblem_objective" ? InfeasSubproblem::SUBPROBLEM_OBJECTIVE
                                                  : InfeasSubproblem::LINEAR_OBJECTIVE;
    const int cf = cn == "subproblem_constraint" ? InfeasSubproblem::SUBPROBLEM_CONSTRAINT
                                                 : InfeasSubproblem::LINEAR_CONSTRAINT;
    infeas = new InfeasSubproblem(sub, of, cf);
  }
  history.clear();
  PO_TRY(sub->initModelAndBounds(tr_size));  // initialize() :1086-1099
  iter_count = 0;
  std::vector<double> con_infeas(m > 0 ? m : 1), model_con_infeas(m > 0 ? m : 1), best_con_infeas;
  for (int i = 0; i < max_it; i++) {
    if (adaptive) PO_TRY(minimizeInfeas(&best_con_infeas));
    if (wfreq > 0 && i % wfreq == 0) sub->writeOutput(i, sub->xk);
    if (iter_cb) iter_cb(iter_cb_user, i);
    PO_TRY(ip->resetDesignAndBounds());
    int rc = ip->optimize(nullptr);
    if (rc != 0 && rc != 1) return rc;
    captureSolveLine(1);
    Vec *step = nullptr, *zw = nullptr;
    const double *z = nullptr;
    ip->getOptimizedPoint(&step, &z, nullptr, nullptr);
    Vec *wv[5];
    ip->getOptimizedSparse(wv);
    zw = wv[0];
    ip->getIterationCounters(&subproblem_iters, nullptr, nullptr);
    if (adaptive) {
      double f0 = 0.0, fm = 0.0;
      if (sub->evalObjCon(nullptr, &f0, con_infeas.data()) != 0) return PO_ERR_USER;
      if (sub->evalObjCon(step, &fm, model_con_infeas.data()) != 0) return PO_ERR_USER;
      for (int j = 0; j < m; j++) {
        con_infeas[j] = j < nineq ? std::max(0.0, -con_infeas[j]) : fabs(con_infeas[j]);
        model_con_infeas[j] = j < nineq ? std::max(0.0, -model_con_infeas[j]) : fabs(model_con_infeas[j]);
      }
    }
    double infeas_v = 0.0, l1 = 0.0, linfty = 0.0;
    PO_TRY(sl1qpUpdate(step, z, zw, &infeas_v, &l1, &linfty));
    if (infeas_v < infeas_tol && (l1 < l1_tol || linfty < linf_tol)) break;
    if (adaptive) {  // :1600-1662
      for (int j = 0; j < m; j++) {
        const double infeas_reduction = con_infeas[j] - model_con_infeas[j];
        const double best_reduction = con_infeas[j] - best_con_infeas[j];
        if (fabs(z[j]) > infeas_tol && con_infeas[j] < infeas_tol && penalty_gamma[j] >= 2.0 * z[j]) {
          penalty_gamma[j] = std::max(0.5 * (penalty_gamma[j] + fabs(z[j])), gmin);
        } else if (con_infeas[j] > infeas_tol && 0.995 * best_reduction > infeas_reduction) {
          penalty_gamma[j] = std::min(1.5 * penalty_gamma[j], gmax);
        }
      }
    }
  }
  flushHistory();
  return 0;
}

void TrustRegion::flushHistory() {
  const std::string fname = options().str("tr_output_file");
  if (ctx->rank != 0 || fname.empty()) return;
  FILE *fp = fopen(fname.c_str(), "w");
  if (!fp) return;
  fputs("ParOptTrustRegion (paropt_amd, MI355X)\n", fp);
  fputs(history.c_str(), fp);
  fclose(fp);
}

}  // namespace po
