// C ABI of the trust-region layer (include/paropt_amd.h, "ParOptTrustRegion").
#include <string.h>

#include "mma.hpp"
#include "tr.hpp"

using namespace po;

#define PO_CHECK_PTR(p)                         \
  do {                                          \
    if (!(p)) {                                 \
      po::set_error("null argument: %s", #p);   \
      return PO_ERR_ARG;                        \
    }                                           \
  } while (0)

namespace {
struct EigCallback {  // C callback behind EigenModelUpdate
  po_eig_update_fn fn;
  void *user;
};
struct EigSynthetic {
  uint64_t seed;
  double curv;
  int64_t offset;
  bool filled;
};
// what the callback wrote through po_vec_get_array pointers of the approximation's vectors (the reference's callback
// fills hvecs through getArray, examples/eigenvalue/eigenvalue_opt.py:157-190) is uploaded when it returns; the
// kernels of the model evaluations read the device copies
int release_mirror(Vec *v, bool upload) {
  if (!v) return PO_OK;
  if (v->h_live && upload && v->h && v->n > 0) {
    PO_HIP(hipMemcpyAsync(v->d, v->h, sizeof(double) * (size_t)v->n, hipMemcpyHostToDevice, v->ctx->stream));
    PO_HIP(hipStreamSynchronize(v->ctx->stream));
  }
  v->h_live = 0;
  return PO_OK;
}
int eig_tramp(void *user, Vec *x, CompactEigenApprox *e) {
  EigCallback *cb = static_cast<EigCallback *>(user);
  po_eig_s h;
  h.e = e;
  const int fail = cb->fn(cb->user, static_cast<po_vec>(x), &h);
  PO_TRY(release_mirror(x, false));
  PO_TRY(release_mirror(e->g0, true));
  for (Vec *v : e->hvecs) PO_TRY(release_mirror(v, true));
  return fail;
}
int eig_synthetic(void *user, Vec *, CompactEigenApprox *e) {
  EigSynthetic *sy = static_cast<EigSynthetic *>(user);
  const int N = e->N;
  if (!sy->filled) {
    for (int i = 0; i < N; i++) {
      PO_TRY(k_fill_hash(e->ctx, e->hvecs[i]->d, e->n, sy->seed, 300 + i, sy->offset, 2.0, -1.0));
      double nrm2 = 0.0;
      PO_TRY(k_reduce1(e->ctx, RED_SUMSQ, e->hvecs[i]->d, nullptr, e->n, &nrm2));
      PO_TRY(k_scale(e->ctx, e->hvecs[i]->d, e->n, 1.0 / sqrt(nrm2)));
    }
    sy->filled = true;
  }
  for (int i = 0; i < N; i++) {
    for (int j = 0; j < N; j++) {
      const double d = -sy->curv * (1.0 + 0.1 * i);
      e->M[(size_t)i * N + j] = i == j ? d : 0.0;
      e->Minv[(size_t)i * N + j] = i == j ? 1.0 / d : 0.0;
    }
  }
  return 0;
}
}  // namespace

struct po_tr_holder {
  EigCallback cb;
  EigSynthetic sy;
};
static po_tr_holder *holder_of(po_tr tr) {
  if (!tr->eig_holder) tr->eig_holder = new po_tr_holder();
  return static_cast<po_tr_holder *>(tr->eig_holder);
}

extern "C" {

int po_tr_create(po_problem prob, po_tr *out) {
  PO_CHECK_PTR(prob);
  PO_CHECK_PTR(out);
  po_tr_s *h = new po_tr_s();
  h->eig_holder = nullptr;
  h->tr = new TrustRegion(prob->p);
  *out = h;
  return PO_OK;
}
int po_tr_destroy(po_tr tr) {
  if (!tr) return PO_OK;
  delete tr->tr;
  delete static_cast<po_tr_holder *>(tr->eig_holder);
  delete tr;
  return PO_OK;
}
int po_tr_set_option_str(po_tr tr, const char *name, const char *value) {
  PO_CHECK_PTR(tr);
  PO_CHECK_PTR(name);
  return tr->tr->options().set(name, value);
}
int po_tr_set_option_int(po_tr tr, const char *name, int value) {
  PO_CHECK_PTR(tr);
  PO_CHECK_PTR(name);
  return tr->tr->options().set(name, value);
}
int po_tr_set_option_float(po_tr tr, const char *name, double value) {
  PO_CHECK_PTR(tr);
  PO_CHECK_PTR(name);
  return tr->tr->options().set(name, value);
}
int po_tr_set_eigen_model(po_tr tr, int N, int index, po_eig_update_fn update, void *user) {
  PO_CHECK_PTR(tr);
  PO_CHECK_PTR(update);
  po_tr_holder *h = holder_of(tr);
  h->cb.fn = update;
  h->cb.user = user;
  return tr->tr->setEigenModel(N, index, &eig_tramp, &h->cb);
}
int po_tr_set_eigen_model_synthetic(po_tr tr, int N, int index, uint64_t seed, double curv) {
  PO_CHECK_PTR(tr);
  if (curv == 0.0) {
    po::set_error("synthetic eigenvalue model needs a non-zero curvature");
    return PO_ERR_ARG;
  }
  po_tr_holder *h = holder_of(tr);
  h->sy.seed = seed;
  h->sy.curv = curv;
  h->sy.offset = tr->tr->prob->offset;
  h->sy.filled = false;
  return tr->tr->setEigenModel(N, index, &eig_synthetic, &h->sy);
}
int po_eig_get_approximation(po_eig approx, double **c0, po_vec *g0, int *N, double **M, double **Minv,
                             const po_vec **hvecs) {
  PO_CHECK_PTR(approx);
  CompactEigenApprox *e = approx->e;
  if (c0) *c0 = &e->c0;
  if (g0) *g0 = static_cast<po_vec>(e->g0);
  if (N) *N = e->N;
  if (M) *M = e->M.data();
  if (Minv) *Minv = e->Minv.data();
  if (hvecs) *hvecs = e->hhandles.data();
  return PO_OK;
}
int po_tr_optimize(po_tr tr) {
  PO_CHECK_PTR(tr);
  return tr->tr->optimize();
}
int po_tr_get_optimized_point(po_tr tr, po_vec *x, const double **z, po_vec *zw) {
  PO_CHECK_PTR(tr);
  TrustRegion *t = tr->tr;
  PO_TRY(t->build());
  if (x) *x = static_cast<po_vec>(t->sub->xk);
  Vec *dummy = nullptr;
  if (z) t->ip->getOptimizedPoint(&dummy, z, nullptr, nullptr);
  if (zw) {
    Vec *wv[5];
    t->ip->getOptimizedSparse(wv);
    *zw = static_cast<po_vec>(wv[0]);
  }
  return PO_OK;
}
int po_tr_get_state(po_tr tr, double *tr_size, int *iter_count, int *subproblem_iters,
                    int *adaptive_subproblem_iters, const double **penalty_gamma, double *fk,
                    const double **ck) {
  PO_CHECK_PTR(tr);
  TrustRegion *t = tr->tr;
  PO_TRY(t->build());
  if (tr_size) *tr_size = t->tr_size;
  if (iter_count) *iter_count = t->iter_count;
  if (subproblem_iters) *subproblem_iters = t->subproblem_iters;
  if (adaptive_subproblem_iters) *adaptive_subproblem_iters = t->adaptive_subproblem_iters;
  if (penalty_gamma) *penalty_gamma = t->penalty_gamma.data();
  if (fk) *fk = t->sub->fk;
  if (ck) *ck = t->sub->ck.data();
  return PO_OK;
}
int po_tr_get_last_row(po_tr tr, const double **row12, const char **info) {
  PO_CHECK_PTR(tr);
  if (row12) *row12 = tr->tr->row;
  if (info) *info = tr->tr->row_info.c_str();
  return PO_OK;
}
int po_tr_get_last_solve_lines(po_tr tr, const char **steering, const char **qp) {
  PO_CHECK_PTR(tr);
  if (steering) *steering = tr->tr->last_solve_line[0].c_str();
  if (qp) *qp = tr->tr->last_solve_line[1].c_str();
  return PO_OK;
}
int po_tr_get_history(po_tr tr, const char **text) {
  PO_CHECK_PTR(tr);
  PO_CHECK_PTR(text);
  *text = tr->tr->history.c_str();
  return PO_OK;
}
int po_tr_get_quasi_newton(po_tr tr, po_qn *qn) {
  PO_CHECK_PTR(tr);
  PO_CHECK_PTR(qn);
  PO_TRY(tr->tr->build());
  *qn = tr->tr->qn_handle.qn ? &tr->tr->qn_handle : nullptr;
  return PO_OK;
}
int po_tr_get_model_vectors(po_tr tr, po_vec *xk, po_vec *gk) {
  PO_CHECK_PTR(tr);
  PO_TRY(tr->tr->build());
  if (xk) *xk = static_cast<po_vec>(tr->tr->sub->xk);
  if (gk) *gk = static_cast<po_vec>(tr->tr->sub->gk);
  return PO_OK;
}
int po_tr_set_iteration_callback(po_tr tr, po_tr_iteration_fn fn, void *user) {
  PO_CHECK_PTR(tr);
  tr->tr->iter_cb = fn;
  tr->tr->iter_cb_user = user;
  return PO_OK;
}

// ---- the trust-region layer piece by piece (the reference's own assembly) -------------------------------------
int po_eig_create(po_problem prob, int N, po_eig *out) {
  PO_CHECK_PTR(prob);
  PO_CHECK_PTR(out);
  if (N < 1 || N > 64) {
    po::set_error("ParOptCompactEigenApprox: N = %d outside 1..64", N);
    return PO_ERR_ARG;
  }
  po_eig_s *h = new po_eig_s();
  h->e = new CompactEigenApprox(prob->p->ctx, prob->p->nlocal, N);
  const int rc = h->e->allocate();
  if (rc != PO_OK) {
    delete h->e;
    delete h;
    return rc;
  }
  *out = h;
  return PO_OK;
}
int po_eig_destroy(po_eig approx) {
  if (!approx) return PO_OK;
  delete approx->e;
  delete approx;
  return PO_OK;
}
// tmp = M (H^T x): the N x N product of multAdd / evalApproximationGradient (.cpp:52-64, 108-120)
static int eig_scaled_dots(CompactEigenApprox *e, Vec *x, std::vector<double> *dots, std::vector<double> *scaled) {
  const int N = e->N;
  std::vector<const double *> hp = e->hPointers();
  dots->assign(N, 0.0);
  PO_TRY(k_mdot(e->ctx, x->d, hp.data(), N, e->n, dots->data()));
  scaled->assign(N, 0.0);
  for (int i = 0; i < N; i++) {
    double v = 0.0;
    for (int j = 0; j < N; j++) v += e->M[(size_t)i * N + j] * (*dots)[j];
    (*scaled)[i] = v;
  }
  return PO_OK;
}
int po_eig_mult_add(po_eig approx, double alpha, po_vec x, po_vec y) {
  PO_CHECK_PTR(approx);
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(y);
  CompactEigenApprox *e = approx->e;
  std::vector<double> dots, sc;
  PO_TRY(eig_scaled_dots(e, x, &dots, &sc));
  for (double &v : sc) v *= alpha;
  std::vector<const double *> hp = e->hPointers();
  return k_panel_axpy(e->ctx, y->d, 0.0, nullptr, 1.0, sc.data(), hp.data(), e->N, e->n);
}
int po_eig_eval_approximation(po_eig approx, po_vec s, po_vec t, double *value) {
  PO_CHECK_PTR(approx);
  PO_CHECK_PTR(value);
  CompactEigenApprox *e = approx->e;
  double c = e->c0;
  if (s && t) {  // the reference only looks at s (t is a scratch argument there too)
    const int N = e->N;
    std::vector<const double *> P = e->hPointers();
    P.push_back(e->g0->d);
    std::vector<double> dots(N + 1, 0.0);
    PO_TRY(k_mdot(e->ctx, s->d, P.data(), N + 1, e->n, dots.data()));
    c += dots[N];
    for (int i = 0; i < N; i++)
      for (int j = 0; j < N; j++) c += 0.5 * e->M[(size_t)i * N + j] * dots[i] * dots[j];
  }
  *value = c;
  return PO_OK;
}
int po_eig_eval_approximation_gradient(po_eig approx, po_vec s, po_vec grad) {
  PO_CHECK_PTR(approx);
  PO_CHECK_PTR(s);
  PO_CHECK_PTR(grad);
  CompactEigenApprox *e = approx->e;
  std::vector<double> dots, sc;
  PO_TRY(eig_scaled_dots(e, s, &dots, &sc));
  std::vector<const double *> hp = e->hPointers();
  return k_panel_axpy(e->ctx, grad->d, 1.0, e->g0->d, 0.0, sc.data(), hp.data(), e->N, e->n);
}
int po_eigqn_create(po_qn qn, po_eig approx, int index, po_qn *out) {
  PO_CHECK_PTR(approx);
  PO_CHECK_PTR(out);
  if (index < 0) {
    po::set_error("ParOptEigenQuasiNewton: negative constraint index");
    return PO_ERR_ARG;
  }
  if (qn && (qn->qn->n != approx->e->n || qn->qn->ctx != approx->e->ctx)) {
    po::set_error("ParOptEigenQuasiNewton: the quasi-Newton object and the approximation differ in size or context");
    return PO_ERR_ARG;
  }
  po_qn_s *h = new po_qn_s();
  h->qn = new EigenQuasiNewton(qn ? qn->qn : nullptr, approx->e, index);
  *out = h;
  return PO_OK;
}
static EigenQuasiNewton *as_eigqn(po_qn q) {
  EigenQuasiNewton *e = q ? dynamic_cast<EigenQuasiNewton *>(q->qn) : nullptr;
  if (!e) po::set_error("not a ParOptEigenQuasiNewton handle");
  return e;
}
int po_eigqn_set_use_quasi_newton_objective(po_qn eig_qn, int truth) {
  EigenQuasiNewton *e = as_eigqn(eig_qn);
  if (!e) return PO_ERR_ARG;
  e->use_qn_objective = truth ? 1 : 0;
  return PO_OK;
}
int po_eigqn_update_multipliers(po_qn eig_qn, const double *z) {
  EigenQuasiNewton *e = as_eigqn(eig_qn);
  if (!e) return PO_ERR_ARG;
  PO_CHECK_PTR(z);
  return e->updateMult(nullptr, z, nullptr);
}
int po_eigqn_get_multiplier_index(po_qn eig_qn, int *index) {
  EigenQuasiNewton *e = as_eigqn(eig_qn);
  if (!e) return PO_ERR_ARG;
  PO_CHECK_PTR(index);
  *index = e->index;
  return PO_OK;
}

struct po_trsub_s {
  TrustRegionSubproblem *sub;
  po_problem_s face;  // the subproblem as a po_problem (po_ip_create, po_problem_eval_*)
  po_qn_s qnh;        // getQuasiNewton(): borrowed handle
  EigCallback cb;
  std::vector<po_vec> akh;
};
static int finish_trsub(po_trsub_s *h, po_trsub *out) {
  const int rc = h->sub->allocate();
  if (rc != PO_OK) {
    delete h->sub;
    delete h;
    return rc;
  }
  h->face.p = h->sub;
  h->qnh.qn = h->sub->getQuasiNewton();
  h->cb.fn = nullptr;
  h->cb.user = nullptr;
  *out = h;
  return PO_OK;
}
int po_trsub_create_quadratic(po_problem prob, po_qn qn, po_trsub *out) {
  PO_CHECK_PTR(prob);
  PO_CHECK_PTR(out);
  if (qn && (qn->qn->n != prob->p->nlocal || qn->qn->ctx != prob->p->ctx)) {
    po::set_error("ParOptQuadraticSubproblem: the quasi-Newton object does not match the problem");
    return PO_ERR_ARG;
  }
  po_trsub_s *h = new po_trsub_s();
  h->sub = new QuadraticSubproblem(prob->p, qn ? qn->qn : nullptr);
  return finish_trsub(h, out);
}
int po_trsub_create_eigen(po_problem prob, po_qn eig_qn, po_trsub *out) {
  PO_CHECK_PTR(prob);
  PO_CHECK_PTR(out);
  EigenQuasiNewton *e = as_eigqn(eig_qn);
  if (!e) return PO_ERR_ARG;
  if (e->n != prob->p->nlocal || e->ctx != prob->p->ctx || e->index >= prob->p->ncon) {
    po::set_error("ParOptEigenSubproblem: the approximation does not match the problem (index %d, ncon %d)", e->index,
                  prob->p->ncon);
    return PO_ERR_ARG;
  }
  po_trsub_s *h = new po_trsub_s();
  h->sub = new EigenSubproblem(prob->p, e);
  return finish_trsub(h, out);
}
int po_trsub_create_callbacks(po_problem prob, const po_trsub_callbacks *callbacks, po_trsub *out) {
  PO_CHECK_PTR(prob);
  PO_CHECK_PTR(callbacks);
  PO_CHECK_PTR(out);
  if (!callbacks->init_model_and_bounds || !callbacks->set_trust_region_bounds ||
      !callbacks->eval_trial_step_and_update || !callbacks->accept_trial_step || !callbacks->get_linear_model ||
      !callbacks->get_vars_and_bounds || !callbacks->eval_obj_con || !callbacks->eval_obj_con_gradient) {
    po::set_error("po_trsub_create_callbacks: a required callback is missing (only get_quasi_newton, "
                  "reject_trial_step and get_quasi_newton_update_type are optional)");
    return PO_ERR_ARG;
  }
  po_trsub_s *h = new po_trsub_s();
  CallbackSubproblem *cs = new CallbackSubproblem(prob->p, *callbacks);
  h->sub = cs;
  const int rc = cs->allocateModel();
  if (rc != PO_OK) {
    delete cs;
    delete h;
    return rc;
  }
  h->face.p = h->sub;
  h->qnh.qn = nullptr;  // asked for when needed: the user's object may not exist yet
  *out = h;
  return PO_OK;
}
int po_trsub_destroy(po_trsub sub) {
  if (!sub) return PO_OK;
  delete sub->sub;
  delete sub;
  return PO_OK;
}
int po_trsub_set_eigen_model_update(po_trsub sub, po_eig_update_fn update, void *user) {
  PO_CHECK_PTR(sub);
  EigenSubproblem *es = dynamic_cast<EigenSubproblem *>(sub->sub);
  if (!es) {
    po::set_error("setEigenModelUpdate: not a ParOptEigenSubproblem");
    return PO_ERR_ARG;
  }
  sub->cb.fn = update;
  sub->cb.user = user;
  es->update_model = update ? &eig_tramp : nullptr;
  es->update_user = &sub->cb;
  return PO_OK;
}
int po_trsub_problem(po_trsub sub, po_problem *out) {
  PO_CHECK_PTR(sub);
  PO_CHECK_PTR(out);
  *out = &sub->face;
  return PO_OK;
}
int po_trsub_sync_linear_model(po_trsub sub) {
  PO_CHECK_PTR(sub);
  CallbackSubproblem *cs = dynamic_cast<CallbackSubproblem *>(sub->sub);
  return cs ? cs->syncLinearModel() : PO_OK;
}
int po_infeas_create(po_trsub sub, int subproblem_objective, int subproblem_constraint, po_problem *out) {
  PO_CHECK_PTR(sub);
  PO_CHECK_PTR(out);
  // the reference's selectors (src/ParOptTrustRegion.h:296-302) -> the library's
  int obj = -1, con = -1;
  if (subproblem_objective == 1) obj = InfeasSubproblem::SUBPROBLEM_OBJECTIVE;
  if (subproblem_objective == 2) obj = InfeasSubproblem::LINEAR_OBJECTIVE;
  if (subproblem_objective == 3) obj = InfeasSubproblem::CONSTANT_OBJECTIVE;
  if (subproblem_constraint == 1) con = InfeasSubproblem::SUBPROBLEM_CONSTRAINT;
  if (subproblem_constraint == 2) con = InfeasSubproblem::LINEAR_CONSTRAINT;
  if (obj < 0 || con < 0) {
    po::set_error("ParOptInfeasSubproblem: objective selector %d (1 subproblem, 2 linear, 3 constant) / constraint "
                  "selector %d (1 subproblem, 2 linear)", subproblem_objective, subproblem_constraint);
    return PO_ERR_ARG;
  }
  // (a user-written subproblem lends its model vectors: taken as they are now)
  if (CallbackSubproblem *cs = dynamic_cast<CallbackSubproblem *>(sub->sub)) PO_TRY(cs->syncLinearModel());
  po_problem_s *h = new po_problem_s();
  h->p = new InfeasSubproblem(sub->sub, obj, con);
  *out = h;
  return PO_OK;
}
int po_infeas_set_objective_scaling(po_problem infeas, double scale) {
  PO_CHECK_PTR(infeas);
  InfeasSubproblem *p = dynamic_cast<InfeasSubproblem *>(infeas->p);
  if (!p) {
    po::set_error("setObjectiveScaling: not a ParOptInfeasSubproblem");
    return PO_ERR_ARG;
  }
  p->obj_scale = scale;
  return PO_OK;
}
int po_trsub_get_quasi_newton(po_trsub sub, po_qn *qn) {
  PO_CHECK_PTR(sub);
  PO_CHECK_PTR(qn);
  sub->qnh.qn = sub->sub->getQuasiNewton();
  *qn = sub->qnh.qn ? &sub->qnh : nullptr;
  return PO_OK;
}
int po_trsub_init_model_and_bounds(po_trsub sub, double tr_size) {
  PO_CHECK_PTR(sub);
  return sub->sub->initModelAndBounds(tr_size);
}
int po_trsub_set_trust_region_bounds(po_trsub sub, double tr_size) {
  PO_CHECK_PTR(sub);
  return sub->sub->setTrustRegionBounds(tr_size);
}
int po_trsub_eval_trial_step_and_update(po_trsub sub, int update_flag, po_vec step, const double *z, po_vec zw,
                                        double *fobj, double *cons) {
  PO_CHECK_PTR(sub);
  PO_CHECK_PTR(step);
  PO_CHECK_PTR(fobj);
  PO_CHECK_PTR(cons);
  return sub->sub->evalTrialStepAndUpdate(update_flag, step, z, zw, fobj, cons);
}
int po_trsub_accept_trial_step(po_trsub sub, po_vec step, const double *z, po_vec zw) {
  PO_CHECK_PTR(sub);
  PO_CHECK_PTR(step);
  return sub->sub->acceptTrialStep(step, z, zw);
}
int po_trsub_reject_trial_step(po_trsub sub) {
  PO_CHECK_PTR(sub);
  sub->sub->rejectTrialStep();
  return PO_OK;
}
int po_trsub_get_quasi_newton_update_type(po_trsub sub, int *type) {
  PO_CHECK_PTR(sub);
  PO_CHECK_PTR(type);
  *type = sub->sub->getQuasiNewtonUpdateType();
  return PO_OK;
}
int po_trsub_get_linear_model(po_trsub sub, po_vec *xk, double *fk, po_vec *gk, const double **ck,
                              const po_vec **Ak, po_vec *lb, po_vec *ub, int *m) {
  PO_CHECK_PTR(sub);
  TrustRegionSubproblem *s = sub->sub;
  if (xk) *xk = static_cast<po_vec>(s->xk);
  if (fk) *fk = s->fk;
  if (gk) *gk = static_cast<po_vec>(s->gk);
  if (ck) *ck = s->ck.data();
  if (Ak) {
    sub->akh.clear();
    for (Vec *a : s->Ak) sub->akh.push_back(static_cast<po_vec>(a));
    *Ak = sub->akh.data();
  }
  if (lb) *lb = static_cast<po_vec>(s->lb);
  if (ub) *ub = static_cast<po_vec>(s->ub);
  if (m) *m = s->m;
  return PO_OK;
}
int po_tr_create_subproblem(po_trsub sub, po_tr *out) {
  PO_CHECK_PTR(sub);
  PO_CHECK_PTR(out);
  po_tr_s *h = new po_tr_s();
  h->eig_holder = nullptr;
  h->tr = new TrustRegion(sub->sub);
  *out = h;
  return PO_OK;
}
int po_tr_optimize_with(po_tr tr, po_ip ip) {
  PO_CHECK_PTR(tr);
  PO_CHECK_PTR(ip);
  return tr->tr->optimize(ip->ip);
}
int po_tr_initialize(po_tr tr) {
  PO_CHECK_PTR(tr);
  return tr->tr->initialize();
}
int po_tr_set_penalty_gamma(po_tr tr, double gamma) {
  PO_CHECK_PTR(tr);
  tr->tr->setPenaltyGamma(gamma);
  return PO_OK;
}
int po_tr_set_penalty_gamma_array(po_tr tr, const double *gamma) {
  PO_CHECK_PTR(tr);
  PO_CHECK_PTR(gamma);
  tr->tr->setPenaltyGammaArray(gamma);
  return PO_OK;
}

// ---- MMA -----------------------------------------------------------------------------------------
int po_mma_create(po_problem prob, po_mma *out) {
  PO_CHECK_PTR(prob);
  PO_CHECK_PTR(out);
  po_mma_s *h = new po_mma_s();
  h->mma = new MMA(prob->p);
  *out = h;
  return PO_OK;
}
int po_mma_destroy(po_mma mma) {
  if (!mma) return PO_OK;
  delete mma->mma;
  delete mma;
  return PO_OK;
}
int po_mma_set_option_str(po_mma mma, const char *name, const char *value) {
  PO_CHECK_PTR(mma);
  PO_CHECK_PTR(name);
  return mma->mma->options().set(name, value);
}
int po_mma_set_option_int(po_mma mma, const char *name, int value) {
  PO_CHECK_PTR(mma);
  PO_CHECK_PTR(name);
  return mma->mma->options().set(name, value);
}
int po_mma_set_option_float(po_mma mma, const char *name, double value) {
  PO_CHECK_PTR(mma);
  PO_CHECK_PTR(name);
  return mma->mma->options().set(name, value);
}
int po_mma_optimize(po_mma mma) {
  PO_CHECK_PTR(mma);
  return mma->mma->optimize();
}
int po_mma_get_optimized_point(po_mma mma, po_vec *x, const double **z, po_vec *zw, po_vec *zl, po_vec *zu) {
  PO_CHECK_PTR(mma);
  MMA *m = mma->mma;
  PO_TRY(m->build());
  if (x) *x = static_cast<po_vec>(m->xvec);
  if (z) *z = m->z.data();
  if (zw) *zw = static_cast<po_vec>(m->zwvec);
  if (zl) *zl = static_cast<po_vec>(m->zlvec);
  if (zu) *zu = static_cast<po_vec>(m->zuvec);
  return PO_OK;
}
int po_mma_get_asymptotes(po_mma mma, po_vec *L, po_vec *U) {
  PO_CHECK_PTR(mma);
  PO_TRY(mma->mma->build());
  if (L) *L = static_cast<po_vec>(mma->mma->Lvec);
  if (U) *U = static_cast<po_vec>(mma->mma->Uvec);
  return PO_OK;
}
int po_mma_get_state(po_mma mma, int *mma_iter, int *subproblem_iter, double *fobj, const double **cons) {
  PO_CHECK_PTR(mma);
  MMA *m = mma->mma;
  if (mma_iter) *mma_iter = m->mma_iter;
  if (subproblem_iter) *subproblem_iter = m->subproblem_iter;
  if (fobj) *fobj = m->fobj;
  if (cons) *cons = m->cons.data();
  return PO_OK;
}
int po_mma_get_last_row(po_mma mma, const double **row5) {
  PO_CHECK_PTR(mma);
  PO_CHECK_PTR(row5);
  *row5 = mma->mma->last_row;
  return PO_OK;
}
int po_mma_get_history(po_mma mma, const char **text) {
  PO_CHECK_PTR(mma);
  PO_CHECK_PTR(text);
  *text = mma->mma->history.c_str();
  return PO_OK;
}
int po_mma_set_iteration_callback(po_mma mma, po_mma_iteration_fn fn, void *user) {
  PO_CHECK_PTR(mma);
  mma->mma->iter_cb = fn;
  mma->mma->iter_cb_user = user;
  return PO_OK;
}

}  // extern "C"
