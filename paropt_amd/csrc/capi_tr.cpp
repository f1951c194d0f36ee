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
int eig_tramp(void *user, Vec *x, CompactEigenApprox *e) {
  EigCallback *cb = static_cast<EigCallback *>(user);
  po_eig_s h;
  h.e = e;
  return cb->fn(cb->user, static_cast<po_vec>(x), &h);
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
