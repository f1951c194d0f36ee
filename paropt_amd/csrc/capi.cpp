// extern "C" surface of libparopt_amd.so (include/paropt_amd.h).
#include <string.h>

#include "ip.hpp"
#include "problem.hpp"
#include "qn.hpp"

namespace po {
const char *last_error();
int ctx_create(int device, Ctx **out);
int ctx_destroy(Ctx *c);
int rccl_unique_id(void *id128);
int comm_init_rccl(Ctx *c, int rank, int size, const void *id128);
int comm_bench(Ctx *c, int count, int pure_sum, int reps, double out_us[3]);
int comm_allreduce_host(Ctx *c, double *values, int count, int op);
int rccl_version();
int comm_init_callback(Ctx *c, int rank, int size, po_allgather_fn fn, void *user);
}  // namespace po

using namespace po;

#define PO_CHECK_PTR(p)                         \
  do {                                          \
    if (!(p)) {                                 \
      po::set_error("null argument: %s", #p);   \
      return PO_ERR_ARG;                        \
    }                                           \
  } while (0)

static int same_layout(po_vec a, po_vec b) {
  if (a->ctx != b->ctx || a->n != b->n) {
    po::set_error("vector mismatch (different context or local size %lld vs %lld)", (long long)a->n,
                  (long long)b->n);
    return PO_ERR_ARG;
  }
  return PO_OK;
}

extern "C" {

const char *po_last_error(void) { return po::last_error(); }
const char *po_version(void) { return "paropt_amd 0.1 gfx950"; }

int po_ctx_create(int device, po_ctx *out) {
  PO_CHECK_PTR(out);
  Ctx *c = nullptr;
  int rc = ctx_create(device, &c);
  *out = static_cast<po_ctx>(c);
  return rc;
}
int po_ctx_destroy(po_ctx ctx) { return ctx_destroy(ctx); }
int po_ctx_synchronize(po_ctx ctx) {
  PO_CHECK_PTR(ctx);
  PO_HIP(hipStreamSynchronize(ctx->stream));
  return PO_OK;
}
int po_ctx_rank(po_ctx ctx, int *rank, int *size) {
  PO_CHECK_PTR(ctx);
  if (rank) *rank = ctx->rank;
  if (size) *size = ctx->size;
  return PO_OK;
}
void *po_ctx_stream(po_ctx ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int po_ctx_counters(po_ctx ctx, int64_t *reductions, int64_t *launches) {
  PO_CHECK_PTR(ctx);
  if (reductions) *reductions = ctx->n_reductions;
  if (launches) *launches = ctx->n_launches;
  return PO_OK;
}
int po_ctx_sync_counters(po_ctx ctx, int64_t *flag_waits, int64_t *flag_timeouts, int64_t *allreduces,
                         int64_t *allgathers) {
  PO_CHECK_PTR(ctx);
  if (flag_waits) *flag_waits = ctx->n_flag_waits;
  if (flag_timeouts) *flag_timeouts = ctx->n_flag_timeouts;
  if (allreduces) *allreduces = ctx->n_allreduce;
  if (allgathers) *allgathers = ctx->n_allgather;
  return PO_OK;
}
int po_ctx_algorithmic_bytes(po_ctx ctx, double *total, double *user) {
  PO_CHECK_PTR(ctx);
  if (total) *total = ctx->alg_bytes;
  if (user) *user = ctx->alg_bytes_user;
  return PO_OK;
}
/* tuning aid (not part of the interface): cycle stamps of the producer/consumer Gram kernel's workgroup 0, see
 * PAROPT_AMD_WGRAM_ABLATE=16 in wgram.hip */
int po_debug_wgram_stamps(double *out8) { return po::wgram_debug_stamps(out8); }
/* tuning aid (not part of the interface): kernel-variant switch for in-process A/B runs, see core.hpp DbgSwitch */
int po_debug_set_switch(int id, int value) {
  po::dbg_switch_set(id, value);
  return PO_OK;
}

int po_ctx_set_reduction_batching(po_ctx ctx, int on) {
  PO_CHECK_PTR(ctx);
  PO_TRY(po::batch_flush(ctx));
  ctx->batch_enabled = on ? 1 : 0;
  return PO_OK;
}
int po_ctx_batched_reductions(po_ctx ctx, int64_t *batched) {
  PO_CHECK_PTR(ctx);
  if (batched) *batched = ctx->n_batched;
  return PO_OK;
}
int po_live_objects(int64_t *vectors, int64_t *bytes) {
  long v = 0;
  long long b = 0;
  po::live_objects(&v, &b);
  if (vectors) *vectors = v;
  if (bytes) *bytes = b;
  return PO_OK;
}
int po_device_count(int *count) {
  PO_CHECK_PTR(count);
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) n = 0;
  *count = n;
  return PO_OK;
}
int po_options_visit_defaults(int which, po_option_visitor fn, void *user) {
  PO_CHECK_PTR(fn);
  po::Options o;  // the interior-point registry
  po::Options base;
  if (which == 1) o.addTrustRegionDefaults();
  if (which == 2) o.addMMADefaults();
  if (which < 0 || which > 2) return PO_ERR_ARG;
  return o.visit(which == 0 ? nullptr : &base, fn, user);
}
int po_live_host_mirrors(int64_t *mirrors) {
  if (mirrors) *mirrors = po::live_mirrors();
  return PO_OK;
}
int po_ctx_time_mdot(po_ctx ctx, int nvecs) {
  PO_CHECK_PTR(ctx);
  ctx->time_mdot_nv = nvecs > 0 ? nvecs : 0;
  ctx->mdot_ms = 0.0;
  ctx->mdot_count = 0;
  return PO_OK;
}
int po_ctx_time_mdot_result(po_ctx ctx, double *ms_total, int64_t *launches) {
  PO_CHECK_PTR(ctx);
  if (ms_total) *ms_total = ctx->mdot_ms;
  if (launches) *launches = ctx->mdot_count;
  return PO_OK;
}
int po_ctx_time_wgram(po_ctx ctx, int on) {
  PO_CHECK_PTR(ctx);
  ctx->time_wgram = on ? 1 : 0;
  for (int i = 0; i < 2; i++) {
    ctx->wgram_ms[i] = 0.0;
    ctx->wgram_count[i] = 0;
    ctx->wgram_cols[i] = 0;
    ctx->wgram_bytes[i] = 0.0;
  }
  return PO_OK;
}
int po_ctx_time_wgram_result(po_ctx ctx, int which, double *ms_total, int64_t *launches, int *ncols,
                             double *alg_bytes_total) {
  PO_CHECK_PTR(ctx);
  if (which < 0 || which > 1) return PO_ERR_ARG;
  if (ms_total) *ms_total = ctx->wgram_ms[which];
  if (launches) *launches = ctx->wgram_count[which];
  if (ncols) *ncols = ctx->wgram_cols[which];
  if (alg_bytes_total) *alg_bytes_total = ctx->wgram_bytes[which];
  return PO_OK;
}
int po_ctx_comm_info(po_ctx ctx, int *kind, int64_t *allreduces, int64_t *allgathers) {
  PO_CHECK_PTR(ctx);
  if (kind) *kind = (int)ctx->comm_kind;
  if (allreduces) *allreduces = ctx->n_allreduce;
  if (allgathers) *allgathers = ctx->n_allgather;
  return PO_OK;
}
int po_ctx_memcpy(po_ctx ctx, void *dst, const void *src, int64_t bytes, int to_device) {
  PO_CHECK_PTR(ctx);
  if (bytes <= 0) return PO_OK;
  PO_CHECK_PTR(dst);
  PO_CHECK_PTR(src);
  PO_HIP(hipSetDevice(ctx->device));
  PO_HIP(hipMemcpyAsync(dst, src, (size_t)bytes, to_device ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost,
                        ctx->stream));
  PO_HIP(hipStreamSynchronize(ctx->stream));
  return PO_OK;
}
int po_ctx_bench_collective(po_ctx ctx, int count, int pure_sum, int reps, double *out_us3) {
  PO_CHECK_PTR(ctx);
  PO_CHECK_PTR(out_us3);
  return comm_bench(ctx, count, pure_sum, reps, out_us3);
}
int po_ctx_allreduce(po_ctx ctx, double *values, int count, int op) {
  PO_CHECK_PTR(ctx);
  if (count > 0) PO_CHECK_PTR(values);
  return comm_allreduce_host(ctx, values, count, op);
}
int po_rccl_version(int *version_code) {
  PO_CHECK_PTR(version_code);
  *version_code = rccl_version();
  return PO_OK;
}
int po_rccl_unique_id(void *id128) {
  PO_CHECK_PTR(id128);
  return rccl_unique_id(id128);
}
int po_ctx_comm_init_rccl(po_ctx ctx, int rank, int size, const void *id128) {
  PO_CHECK_PTR(ctx);
  return comm_init_rccl(ctx, rank, size, id128);
}
int po_ctx_comm_init_callback(po_ctx ctx, int rank, int size, po_allgather_fn fn, void *user) {
  PO_CHECK_PTR(ctx);
  return comm_init_callback(ctx, rank, size, fn, user);
}

// ---- vectors ------------------------------------------------------------------------------------
// getArray contract (src/ParOptVec.cpp:212-217: the pointer IS the data).  While a mirror is live, reads through
// the ABI see what the host wrote and results of ABI operations are visible through the pointer at once.
static int mirror_up(Vec *v) {
  if (!v || !v->h_live || !v->h || v->n <= 0) return PO_OK;
  PO_HIP(hipMemcpyAsync(v->d, v->h, sizeof(double) * (size_t)v->n, hipMemcpyHostToDevice, v->ctx->stream));
  return PO_OK;  // stream-ordered before the kernels that follow
}
static int mirror_down(Vec *v) {
  if (!v || !v->h_live || !v->h || v->n <= 0) return PO_OK;
  PO_HIP(hipMemcpyAsync(v->h, v->d, sizeof(double) * (size_t)v->n, hipMemcpyDeviceToHost, v->ctx->stream));
  PO_HIP(hipStreamSynchronize(v->ctx->stream));
  return PO_OK;
}
int po_vec_create(po_ctx ctx, int64_t nlocal, po_vec *out) {
  PO_CHECK_PTR(ctx);
  PO_CHECK_PTR(out);
  if (nlocal < 0) {
    po::set_error("negative vector length");
    return PO_ERR_ARG;
  }
  Vec *v = vec_new(ctx, nlocal);
  *out = static_cast<po_vec>(v);
  return v ? PO_OK : PO_ERR_HIP;
}
int po_vec_incref(po_vec v) {
  PO_CHECK_PTR(v);
  v->ref++;
  return PO_OK;
}
int po_vec_decref(po_vec v) {
  PO_CHECK_PTR(v);
  vec_decref(v);
  return PO_OK;
}
int po_vec_size(po_vec v, int64_t *nlocal) {
  PO_CHECK_PTR(v);
  *nlocal = v->n;
  return PO_OK;
}
int po_vec_set(po_vec v, double alpha) {
  PO_CHECK_PTR(v);
  PO_TRY(k_fill(v->ctx, v->d, v->n, alpha));
  return mirror_down(v);
}
int po_vec_zero(po_vec v) {
  PO_CHECK_PTR(v);
  PO_TRY(k_fill(v->ctx, v->d, v->n, 0.0));
  return mirror_down(v);
}
int po_vec_copy(po_vec dst, po_vec src) {
  PO_CHECK_PTR(dst);
  PO_CHECK_PTR(src);
  PO_TRY(same_layout(dst, src));
  PO_TRY(mirror_up(src));
  PO_TRY(k_copy(dst->ctx, dst->d, src->d, dst->n));
  return mirror_down(dst);
}
int po_vec_scale(po_vec v, double alpha) {
  PO_CHECK_PTR(v);
  PO_TRY(mirror_up(v));
  PO_TRY(k_scale(v->ctx, v->d, v->n, alpha));
  return mirror_down(v);
}
int po_vec_axpy(po_vec y, double alpha, po_vec x) {
  PO_CHECK_PTR(y);
  PO_CHECK_PTR(x);
  PO_TRY(same_layout(y, x));
  PO_TRY(mirror_up(x));
  PO_TRY(mirror_up(y));
  PO_TRY(k_axpy(y->ctx, y->d, alpha, x->d, y->n));
  return mirror_down(y);
}
int po_vec_dot(po_vec x, po_vec y, double *out) {
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(y);
  PO_CHECK_PTR(out);
  PO_TRY(same_layout(x, y));
  PO_TRY(mirror_up(x));
  PO_TRY(mirror_up(y));
  return k_reduce1(x->ctx, RED_DOT, x->d, y->d, x->n, out);
}
int po_vec_mdot(po_vec x, const po_vec *vecs, int nvecs, double *out) {
  PO_CHECK_PTR(x);
  if (nvecs <= 0) return PO_OK;
  PO_CHECK_PTR(vecs);
  PO_CHECK_PTR(out);
  std::vector<const double *> p(nvecs);
  for (int j = 0; j < nvecs; j++) {
    PO_CHECK_PTR(vecs[j]);
    PO_TRY(same_layout(x, vecs[j]));
    PO_TRY(mirror_up(vecs[j]));
    p[j] = vecs[j]->d;
  }
  PO_TRY(mirror_up(x));
  // wider panels than one kernel accepts are processed in slabs
  int j0 = 0;
  while (j0 < nvecs) {
    const int w = (nvecs - j0 > kMaxPanel) ? kMaxPanel : nvecs - j0;
    PO_TRY(k_mdot(x->ctx, x->d, p.data() + j0, w, x->n, out + j0));
    j0 += w;
  }
  return PO_OK;
}
int po_vec_norm(po_vec x, double *out) {
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(out);
  double ss = 0.0;
  PO_TRY(mirror_up(x));
  PO_TRY(k_reduce1(x->ctx, RED_SUMSQ, x->d, nullptr, x->n, &ss));
  *out = sqrt(ss);
  return PO_OK;
}
int po_vec_maxabs(po_vec x, double *out) {
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(out);
  PO_TRY(mirror_up(x));
  return k_reduce1(x->ctx, RED_AMAX, x->d, nullptr, x->n, out);
}
int po_vec_l1norm(po_vec x, double *out) {
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(out);
  PO_TRY(mirror_up(x));
  return k_reduce1(x->ctx, RED_ASUM, x->d, nullptr, x->n, out);
}
static int ensure_mirror(po_vec v) {
  if (!v->h) {
    PO_HIP(hipHostMalloc((void **)&v->h, sizeof(double) * (size_t)(v->n > 0 ? v->n : 1), hipHostMallocDefault));
    po::mirror_created();
  }
  return PO_OK;
}
int po_vec_get_array(po_vec v, double **host) {
  PO_CHECK_PTR(v);
  PO_CHECK_PTR(host);
  PO_TRY(ensure_mirror(v));
  if (!v->h_live) {  // first hand-out since the last release: bring the mirror up to date, then it is live
    PO_HIP(hipMemcpyAsync(v->h, v->d, sizeof(double) * (size_t)v->n, hipMemcpyDeviceToHost, v->ctx->stream));
    PO_HIP(hipStreamSynchronize(v->ctx->stream));
    v->h_live = 1;
  }
  *host = v->h;
  return PO_OK;
}
int po_vec_peek_array(po_vec v, double **host) {
  PO_CHECK_PTR(v);
  PO_CHECK_PTR(host);
  PO_TRY(ensure_mirror(v));
  if (!v->h_live) {
    PO_HIP(hipMemcpyAsync(v->h, v->d, sizeof(double) * (size_t)v->n, hipMemcpyDeviceToHost, v->ctx->stream));
    PO_HIP(hipStreamSynchronize(v->ctx->stream));
  }
  *host = v->h;
  return PO_OK;
}
int po_vec_release_array(po_vec v, int upload) {
  PO_CHECK_PTR(v);
  if (v->h_live && upload) {
    PO_HIP(hipMemcpyAsync(v->d, v->h, sizeof(double) * (size_t)v->n, hipMemcpyHostToDevice, v->ctx->stream));
    PO_HIP(hipStreamSynchronize(v->ctx->stream));
  }
  v->h_live = 0;
  return PO_OK;
}
int po_vec_sync_to_device(po_vec v) {
  PO_CHECK_PTR(v);
  if (!v->h) return PO_OK;
  PO_HIP(hipMemcpyAsync(v->d, v->h, sizeof(double) * (size_t)v->n, hipMemcpyHostToDevice, v->ctx->stream));
  PO_HIP(hipStreamSynchronize(v->ctx->stream));
  return PO_OK;
}
int po_vec_sync_to_host(po_vec v) {  // a read-only download: does not make the mirror live
  PO_CHECK_PTR(v);
  PO_TRY(ensure_mirror(v));
  PO_HIP(hipMemcpyAsync(v->h, v->d, sizeof(double) * (size_t)v->n, hipMemcpyDeviceToHost, v->ctx->stream));
  PO_HIP(hipStreamSynchronize(v->ctx->stream));
  return PO_OK;
}
int po_vec_get_device_array(po_vec v, double **device) {
  PO_CHECK_PTR(v);
  PO_CHECK_PTR(device);
  *device = v->d;
  return PO_OK;
}
int po_vec_maxpy(po_vec y, double beta, const double *alpha, const po_vec *vecs, int nvecs) {
  PO_CHECK_PTR(y);
  std::vector<const double *> p(nvecs > 0 ? nvecs : 1);
  for (int j = 0; j < nvecs; j++) {
    PO_CHECK_PTR(vecs[j]);
    PO_TRY(same_layout(y, vecs[j]));
    PO_TRY(mirror_up(vecs[j]));
    p[j] = vecs[j]->d;
  }
  PO_TRY(mirror_up(y));
  PO_TRY(k_panel_axpy(y->ctx, y->d, 0.0, nullptr, beta, alpha, p.data(), nvecs, y->n));
  return mirror_down(y);
}
int po_vec_fill_hash(po_vec v, uint64_t seed, uint64_t array_id, int64_t offset, double scale,
                     double shift) {
  PO_CHECK_PTR(v);
  PO_TRY(k_fill_hash(v->ctx, v->d, v->n, seed, array_id, offset, scale, shift));
  return mirror_down(v);
}

// ---- quasi-Newton -------------------------------------------------------------------------------
int po_qn_create(po_ctx ctx, int type, int64_t nlocal, int subspace, po_qn *out) {
  PO_CHECK_PTR(ctx);
  PO_CHECK_PTR(out);
  if (subspace < 0 || nlocal < 0 || (type != PO_QN_BFGS && type != PO_QN_SR1)) {
    po::set_error("bad quasi-Newton arguments");
    return PO_ERR_ARG;
  }
  po_qn_s *h = new po_qn_s();
  if (type == PO_QN_BFGS) {
    h->qn = new LBFGS(ctx, nlocal, subspace);
  } else {
    h->qn = new LSR1(ctx, nlocal, subspace);
  }
  *out = h;
  return PO_OK;
}
int po_qn_destroy(po_qn qn) {
  if (!qn) return PO_OK;
  delete qn->qn;
  delete qn;
  return PO_OK;
}
int po_qn_set_update_type(po_qn qn, int t) {
  PO_CHECK_PTR(qn);
  LBFGS *b = dynamic_cast<LBFGS *>(qn->qn);
  if (b) b->setBFGSUpdateType(t);
  return PO_OK;
}
int po_qn_set_diag_type(po_qn qn, int t) {
  PO_CHECK_PTR(qn);
  qn->qn->setInitDiagonalType(t);
  return PO_OK;
}
int po_qn_reset(po_qn qn) {
  PO_CHECK_PTR(qn);
  qn->qn->reset();
  return PO_OK;
}
int po_qn_update(po_qn qn, po_vec s, po_vec y, int *rc) {
  PO_CHECK_PTR(qn);
  PO_CHECK_PTR(s);
  PO_CHECK_PTR(y);
  int r = 0;
  PO_TRY(mirror_up(s));
  PO_TRY(mirror_up(y));
  PO_TRY(qn->qn->update(s, y, &r));
  if (rc) *rc = r;
  return PO_OK;
}
int po_qn_mult(po_qn qn, po_vec x, po_vec y) {
  PO_CHECK_PTR(qn);
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(y);
  PO_TRY(mirror_up(x));
  PO_TRY(qn->qn->mult(x, y));
  return mirror_down(y);
}
int po_qn_mult_add(po_qn qn, double alpha, po_vec x, po_vec y) {
  PO_CHECK_PTR(qn);
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(y);
  PO_TRY(mirror_up(x));
  PO_TRY(mirror_up(y));
  PO_TRY(qn->qn->multAdd(alpha, x, y));
  return mirror_down(y);
}
int po_qn_get_compact(po_qn qn, int *size, double *b0, const double **d0, const double **M,
                      const po_vec **Z) {
  PO_CHECK_PTR(qn);
  Vec **z = nullptr;
  int k = qn->qn->getCompactMat(b0, d0, M, &z);
  if (size) *size = k;
  if (Z) {
    qn->zhandles.resize(k > 0 ? k : 1);
    for (int i = 0; i < k; i++) qn->zhandles[i] = static_cast<po_vec>(z[i]);
    *Z = qn->zhandles.data();
  }
  return PO_OK;
}
int po_qn_get_pivots(po_qn qn, const int **mfpiv, int *n) {
  PO_CHECK_PTR(qn);
  if (mfpiv) *mfpiv = qn->qn->pivots().data();
  if (n) *n = (int)qn->qn->pivots().size();
  return PO_OK;
}
int po_qn_debug_load(po_qn qn, int msub, double b0, const double *B, const double *L, const double *D, int ld,
                     const po_vec *S, const po_vec *Y) {
  PO_CHECK_PTR(qn);
  if (msub > 0) {
    PO_CHECK_PTR(B);
    PO_CHECK_PTR(L);
    PO_CHECK_PTR(D);
    PO_CHECK_PTR(S);
    PO_CHECK_PTR(Y);
  }
  std::vector<Vec *> s(msub > 0 ? msub : 1, nullptr), y(msub > 0 ? msub : 1, nullptr);
  for (int j = 0; j < msub; j++) {
    s[j] = S[j];
    y[j] = Y[j];
  }
  qn->zhandles.clear();
  return qn->qn->debugLoad(msub, b0, B, L, D, ld, s.data(), y.data());
}
int po_qn_max_size(po_qn qn, int *size) {
  PO_CHECK_PTR(qn);
  *size = qn->qn->getMaxLimitedMemorySize();
  return PO_OK;
}

// ---- problems -------------------------------------------------------------------------------------
int po_problem_create_callbacks(po_ctx ctx, int64_t nlocal, int ncon, int ninequality,
                                const po_problem_callbacks *cb, po_problem *out) {
  PO_CHECK_PTR(ctx);
  PO_CHECK_PTR(cb);
  PO_CHECK_PTR(out);
  if (!cb->get_vars_and_bounds || !cb->eval_obj_con || !cb->eval_obj_con_gradient) {
    po::set_error("get_vars_and_bounds, eval_obj_con and eval_obj_con_gradient are mandatory");
    return PO_ERR_ARG;
  }
  po_problem_s *h = new po_problem_s();
  h->p = new CallbackProblem(ctx, nlocal, ncon, ninequality < 0 ? ncon : ninequality, *cb);
  *out = h;
  return PO_OK;
}
int po_problem_create_separable(po_ctx ctx, int kind, int64_t nglobal, int ncon, uint64_t seed,
                                double eig_min, double eig_max, po_problem *out) {
  PO_CHECK_PTR(ctx);
  PO_CHECK_PTR(out);
  if (kind < 0 || kind > 2 || nglobal < 1 || ncon < 0) {
    po::set_error("bad separable problem arguments");
    return PO_ERR_ARG;
  }
  SeparableProblem *p = new SeparableProblem(ctx, kind, nglobal, ncon, seed, eig_min, eig_max);
  int rc = p->init();
  if (rc != PO_OK) {
    delete p;
    return rc;
  }
  po_problem_s *h = new po_problem_s();
  h->p = p;
  *out = h;
  return PO_OK;
}
int po_problem_set_sparse_callbacks(po_problem p, int64_t nwcon, int64_t nwinequality,
                                    const po_problem_sparse_callbacks *cb) {
  PO_CHECK_PTR(p);
  PO_CHECK_PTR(cb);
  CallbackProblem *q = dynamic_cast<CallbackProblem *>(p->p);
  if (!q) {
    po::set_error("po_problem_set_sparse_callbacks needs a callback problem");
    return PO_ERR_ARG;
  }
  if (nwcon < 0 || nwinequality < 0 || nwinequality > nwcon || !cb->eval_sparse_con ||
      !cb->add_sparse_jacobian || !cb->add_sparse_jacobian_transpose || !cb->add_sparse_inner_product) {
    po::set_error("bad sparse constraint sizes or missing sparse callbacks");
    return PO_ERR_ARG;
  }
  q->sparse.cb = *cb;
  q->sparse.set = true;
  const bool resized = q->nwcon != nwcon;
  q->nwcon = nwcon;
  q->nwinequality = nwinequality;
  // packed-block buffers sized for the previous constraint count are stale: rebuild them for the new one
  if (resized && q->nwblock > 1) {
    const int nb = q->nwblock;
    if (nwcon % nb != 0) {
      q->setSparseBlockSize(1);
      po::set_error("the new sparse constraint count %lld is not a multiple of the block size %d: block size reset "
                    "to 1", (long long)nwcon, nb);
      return PO_ERR_ARG;
    }
    return q->setSparseBlockSize(nb);
  }
  return PO_OK;
}
int po_problem_set_sparse_jacobian_data(po_problem p, int64_t nwcon, int64_t nwinequality, const int *rowp,
                                        const int *cols, po_eval_sparse_obj_con_fn eval_sparse_obj_con,
                                        po_eval_sparse_obj_con_gradient_fn eval_sparse_obj_con_gradient) {
  PO_CHECK_PTR(p);
  PO_CHECK_PTR(rowp);
  CallbackProblem *q = dynamic_cast<CallbackProblem *>(p->p);
  if (!q) {
    po::set_error("po_problem_set_sparse_jacobian_data needs a callback problem");
    return PO_ERR_ARG;
  }
  if (!eval_sparse_obj_con || !eval_sparse_obj_con_gradient) {
    po::set_error("po_problem_set_sparse_jacobian_data: both evaluation callbacks are required");
    return PO_ERR_ARG;
  }
  PO_TRY(q->setSparseJacobianData(nwcon, nwinequality, rowp, cols));
  q->csr_obj_con = eval_sparse_obj_con;
  q->csr_gradient = eval_sparse_obj_con_gradient;
  return PO_OK;
}
int po_problem_get_sparse_jacobian_data(po_problem p, const int **rowp, const int **cols, double **data,
                                        int64_t *nnz) {
  PO_CHECK_PTR(p);
  po::CsrSparse *m = p->p->csr;
  if (!m) {
    po::set_error("the problem has no CSR sparse Jacobian");
    return PO_ERR_ARG;
  }
  if (rowp) *rowp = m->user_rowp.data();
  if (cols) *cols = m->user_cols.data();
  if (data) *data = m->data;
  if (nnz) *nnz = m->nnz;
  return PO_OK;
}
int po_quasidef_factor(po_problem p, po_vec x, po_vec dinv, po_vec c) {
  PO_CHECK_PTR(p);
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(dinv);
  PO_CHECK_PTR(c);
  if (p->p->nwcon <= 0 || c->n != p->p->nwcon || dinv->n != p->p->nlocal) {
    po::set_error("po_quasidef_factor: the problem has no sparse constraints or the sizes do not match");
    return PO_ERR_ARG;
  }
  const long before = p->p->sparseFactorBreakdowns();
  PO_TRY(p->p->sparseFactor(x, dinv, c));
  return p->p->sparseFactorBreakdowns() != before ? PO_ERR_NUMERIC : PO_OK;  // the text is already set
}
int po_quasidef_apply(po_problem p, po_vec x, po_vec dinv, po_vec c, po_vec bx, po_vec bw, po_vec yx,
                      po_vec yw) {
  PO_CHECK_PTR(p);
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(dinv);
  PO_CHECK_PTR(c);
  PO_CHECK_PTR(bx);
  PO_CHECK_PTR(yx);
  PO_CHECK_PTR(yw);
  po::Problem *q = p->p;
  if (q->nwcon <= 0 || c->n != q->nwcon || yw->n != q->nwcon || (bw && bw->n != q->nwcon) ||
      dinv->n != q->nlocal || bx->n != q->nlocal || yx->n != q->nlocal || bx->d == yx->d) {
    po::set_error("po_quasidef_apply: size mismatch (or bx aliases yx)");
    return PO_ERR_ARG;
  }
  po::Vec *work = po::vec_new(q->ctx, q->nwcon);
  if (!work) return PO_ERR_HIP;
  int rc = q->sparseApplyK0(x, dinv, c, bx->d, bw ? bw->d : nullptr, yx, yw, work);
  if (rc == PO_OK && hipStreamSynchronize(q->ctx->stream) != hipSuccess) rc = PO_ERR_HIP;
  po::vec_decref(work);
  return rc;
}
struct po_csr_symbolic_s {
  po::CsrSymbolic s;
};
int po_csr_symbolic_create(int64_t nvars, int64_t nwcon, const int *rowp, const int *cols, po_csr_symbolic *out) {
  PO_CHECK_PTR(rowp);
  PO_CHECK_PTR(out);
  if (nwcon < 0 || nvars < 0 || (rowp[nwcon] > 0 && !cols)) {
    po::set_error("po_csr_symbolic_create: bad arguments");
    return PO_ERR_ARG;
  }
  po_csr_symbolic h = new po_csr_symbolic_s;
  int rc = po::csr_analyse(nvars, nwcon, rowp, cols, &h->s);
  if (rc != PO_OK) {
    delete h;
    return rc;
  }
  *out = h;
  return PO_OK;
}
int po_csr_symbolic_info(po_csr_symbolic h, int64_t info[7]) {
  PO_CHECK_PTR(h);
  PO_CHECK_PTR(info);
  info[0] = h->s.nnz;
  info[1] = h->s.nnzS;
  info[2] = h->s.nnzL;
  info[3] = (int64_t)h->s.fwd_ptr.size() - 1;
  info[4] = h->s.identity_src ? 1 : 0;
  info[5] = (int64_t)h->s.front_start.size();
  info[6] = h->s.max_front;
  return PO_OK;
}
int po_csr_symbolic_arrays(po_csr_symbolic h, const int **perm, const int **parent, const int **Lrowp,
                           const int **Lcols, const int **level_ptr, const int **front_of) {
  PO_CHECK_PTR(h);
  if (perm) *perm = h->s.perm.data();
  if (parent) *parent = h->s.parent.data();
  if (Lrowp) *Lrowp = h->s.Lrowp.data();
  if (Lcols) *Lcols = h->s.Lcols.data();
  if (level_ptr) *level_ptr = h->s.fwd_ptr.data();
  if (front_of) *front_of = h->s.front_of.data();
  return PO_OK;
}
int po_csr_symbolic_destroy(po_csr_symbolic h) {
  delete h;
  return PO_OK;
}
const char *po_quasidef_factor_info(po_problem p) { return p && p->p ? p->p->sparseFactorInfo() : nullptr; }
int po_problem_set_sparse_block_size(po_problem p, int nwblock) {
  PO_CHECK_PTR(p);
  CallbackProblem *q = dynamic_cast<CallbackProblem *>(p->p);
  if (!q || !q->sparse.set) {
    po::set_error("po_problem_set_sparse_block_size needs a callback problem with sparse callbacks");
    return PO_ERR_ARG;
  }
  return q->setSparseBlockSize(nwblock);
}
int po_problem_set_hessian_callbacks(po_problem p, po_hvec_fn hvec, po_hdiag_fn hdiag) {
  PO_CHECK_PTR(p);
  CallbackProblem *q = dynamic_cast<CallbackProblem *>(p->p);
  if (!q) {
    po::set_error("po_problem_set_hessian_callbacks needs a callback problem");
    return PO_ERR_ARG;
  }
  q->hvec_fn = hvec;
  q->hdiag_fn = hdiag;
  return PO_OK;
}
int po_problem_set_weighting(po_problem p, int64_t nwcon, int nw, int64_t nwstart, int nwskip,
                             int64_t nwinequality) {
  PO_CHECK_PTR(p);
  SeparableProblem *q = dynamic_cast<SeparableProblem *>(p->p);
  if (!q) {
    po::set_error("po_problem_set_weighting needs a built-in separable problem");
    return PO_ERR_ARG;
  }
  return q->setWeighting(nwcon, nw, nwstart, nwskip, nwinequality);
}
int po_problem_set_chain(po_problem p, int span, int stride, int reverse_cols) {
  PO_CHECK_PTR(p);
  SeparableProblem *q = dynamic_cast<SeparableProblem *>(p->p);
  if (!q) {
    po::set_error("po_problem_set_chain needs a built-in separable problem");
    return PO_ERR_ARG;
  }
  return q->setChain(span, stride, reverse_cols);
}
int po_problem_sparse_sizes(po_problem p, int64_t *nwcon_local, int64_t *nwinequality_local) {
  PO_CHECK_PTR(p);
  if (nwcon_local) *nwcon_local = p->p->nwcon;
  if (nwinequality_local) *nwinequality_local = p->p->nwinequality;
  return PO_OK;
}
int po_problem_set_var_bound_options(po_problem p, int use_lower, int use_upper) {
  PO_CHECK_PTR(p);
  p->p->use_lower = use_lower ? 1 : 0;
  p->p->use_upper = use_upper ? 1 : 0;
  return PO_OK;
}
int po_problem_set_bounds_mode(po_problem p, int mode) {
  PO_CHECK_PTR(p);
  SeparableProblem *sp = dynamic_cast<SeparableProblem *>(p->p);
  if (!sp || mode < 0 || mode > 7) {
    set_error("po_problem_set_bounds_mode: built-in problems only, mode 0..7");
    return PO_ERR_ARG;
  }
  sp->bounds_mode = mode;
  return PO_OK;
}
int po_problem_set_deferred_reductions(po_problem p, int flag) {
  PO_CHECK_PTR(p);
  po::CallbackProblem *cp = dynamic_cast<po::CallbackProblem *>(p->p);
  if (!cp) {
    po::set_error("po_problem_set_deferred_reductions: not a callback problem (the built-in problems always defer)");
    return PO_ERR_ARG;
  }
  cp->deferred_reductions = flag ? 1 : 0;
  return PO_OK;
}
int po_ctx_after_reduce(po_ctx ctx, po_after_reduce_fn fn, void *user) {
  PO_CHECK_PTR(ctx);
  PO_CHECK_PTR(fn);
  po::after_reduce(ctx, [fn, user] { fn(user); });
  return PO_OK;
}
int po_ctx_reduce_device(po_ctx ctx, const double *device_values, int count, int op, double *host_out) {
  PO_CHECK_PTR(ctx);
  PO_CHECK_PTR(host_out);
  if (count <= 0) return PO_OK;
  PO_CHECK_PTR(device_values);
  if (op < 0 || op > 2 || count > po::kMaxRed) {
    po::set_error("po_ctx_reduce_device: count %d / op %d out of range", count, op);
    return PO_ERR_ARG;
  }
  // one "block" of first-stage partials: the values themselves
  PO_TRY(po::ensure_partials(ctx, (size_t)count));
  PO_HIP(hipMemcpyAsync(ctx->d_partials, device_values, sizeof(double) * (size_t)count, hipMemcpyDeviceToDevice,
                        ctx->stream));
  return po::reduce_finish(ctx, 1, op == 0 ? count : 0, op == 1 ? count : 0, op == 2 ? count : 0, host_out);
}
int po_problem_set_linear_constraints(po_problem p, int flag) {
  PO_CHECK_PTR(p);
  SeparableProblem *sp = dynamic_cast<SeparableProblem *>(p->p);
  if (flag && sp && sp->kind == PO_PROBLEM_ROSENBROCK) {
    set_error("the Rosenbrock problem's constraints are not linear");
    return PO_ERR_ARG;
  }
  p->p->linear_constraints = flag ? 1 : 0;
  return PO_OK;
}
int po_problem_destroy(po_problem p) {
  if (!p) return PO_OK;
  delete p->p;
  delete p;
  return PO_OK;
}
int po_problem_sizes(po_problem p, int64_t *nlocal, int64_t *offset, int *ncon) {
  PO_CHECK_PTR(p);
  if (nlocal) *nlocal = p->p->nlocal;
  if (offset) *offset = p->p->offset;
  if (ncon) *ncon = p->p->ncon;
  return PO_OK;
}
int po_problem_eval_obj_con(po_problem p, po_vec x, double *fobj, double *cons) {
  PO_CHECK_PTR(p);
  PO_CHECK_PTR(x);
  PO_TRY(mirror_up(x));
  return p->p->evalObjCon(x, fobj, cons);
}
int po_problem_eval_obj_con_gradient(po_problem p, po_vec x, po_vec g, const po_vec *Ac) {
  PO_CHECK_PTR(p);
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(g);
  std::vector<Vec *> a(p->p->ncon > 0 ? p->p->ncon : 1);
  for (int j = 0; j < p->p->ncon; j++) a[j] = Ac[j];
  PO_TRY(mirror_up(x));
  int rc = p->p->evalObjConGradient(x, g, a.data());
  PO_TRY(mirror_down(g));
  for (int j = 0; j < p->p->ncon; j++) PO_TRY(mirror_down(a[j]));
  return rc;
}
int po_problem_get_vars_and_bounds(po_problem p, po_vec x, po_vec lb, po_vec ub) {
  PO_CHECK_PTR(p);
  int rc = p->p->getVarsAndBounds(x, lb, ub);
  PO_TRY(mirror_down(x));
  PO_TRY(mirror_down(lb));
  PO_TRY(mirror_down(ub));
  return rc;
}

// ---- interior point -----------------------------------------------------------------------------
int po_ip_create(po_problem prob, po_ip *out) {
  PO_CHECK_PTR(prob);
  PO_CHECK_PTR(out);
  po_ip_s *h = new po_ip_s();
  h->ip = new InteriorPoint(prob->p);
  int rc = h->ip->allocate();
  if (rc != PO_OK) {
    delete h->ip;
    delete h;
    return rc;
  }
  *out = h;
  return PO_OK;
}
int po_ip_destroy(po_ip ip) {
  if (!ip) return PO_OK;
  delete ip->ip;
  delete ip;
  return PO_OK;
}
int po_ip_check_gradients(po_ip ip, double dh, const char **report) {
  PO_CHECK_PTR(ip);
  static thread_local std::string text;
  text.clear();
  PO_TRY(ip->ip->checkGradients(dh, &text));
  if (report) *report = text.c_str();
  return PO_OK;
}
int po_ip_check_merit_func_gradient(po_ip ip, po_vec xpt, double dh, double *fd, double *actual) {
  PO_CHECK_PTR(ip);
  if (xpt) PO_TRY(mirror_up(xpt));
  double out[2] = {0.0, 0.0};
  PO_TRY(ip->ip->checkMeritFuncGradient(xpt, dh, out));
  if (fd) *fd = out[0];
  if (actual) *actual = out[1];
  return PO_OK;
}
int po_ip_set_option_str(po_ip ip, const char *name, const char *value) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(name);
  return ip->ip->options.set(name, value);
}
int po_ip_set_option_int(po_ip ip, const char *name, int value) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(name);
  return ip->ip->options.set(name, value);
}
int po_ip_set_option_float(po_ip ip, const char *name, double value) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(name);
  int rc = ip->ip->options.set(name, value);
  if (rc == PO_OK && strcmp(name, "penalty_gamma") == 0) ip->ip->setPenaltyGamma(value);
  return rc;
}
int po_ip_optimize(po_ip ip, const char *checkpoint) {
  PO_CHECK_PTR(ip);
  return ip->ip->optimize(checkpoint);
}
int po_ip_get_optimized_point(po_ip ip, po_vec *x, const double **z, po_vec *zl, po_vec *zu) {
  PO_CHECK_PTR(ip);
  Vec *vx, *vzl, *vzu;
  ip->ip->getOptimizedPoint(&vx, z, &vzl, &vzu);
  if (x) *x = static_cast<po_vec>(vx);
  if (zl) *zl = static_cast<po_vec>(vzl);
  if (zu) *zu = static_cast<po_vec>(vzu);
  return PO_OK;
}
int po_ip_get_optimized_slacks(po_ip ip, const double **s, const double **t, const double **zs,
                               const double **zt) {
  PO_CHECK_PTR(ip);
  ip->ip->getOptimizedSlacks(s, t, zs, zt);
  return PO_OK;
}
int po_ip_get_optimized_sparse(po_ip ip, po_vec *zw, po_vec *sw, po_vec *tw, po_vec *zsw,
                               po_vec *ztw) {
  PO_CHECK_PTR(ip);
  Vec *v[5];
  ip->ip->getOptimizedSparse(v);
  po_vec *o[5] = {zw, sw, tw, zsw, ztw};
  for (int i = 0; i < 5; i++)
    if (o[i]) *o[i] = static_cast<po_vec>(v[i]);
  return PO_OK;
}
int po_ip_get_counters(po_ip ip, int *niter, int *neval, int *ngeval) {
  PO_CHECK_PTR(ip);
  ip->ip->getIterationCounters(niter, neval, ngeval);
  return PO_OK;
}
int po_ip_get_barrier_parameter(po_ip ip, double *mu) {
  PO_CHECK_PTR(ip);
  *mu = ip->ip->getBarrierParameter();
  return PO_OK;
}
int po_ip_get_complementarity(po_ip ip, double *comp) {
  PO_CHECK_PTR(ip);
  return ip->ip->getComplementarity(comp);
}
int po_ip_get_objective(po_ip ip, double *fobj, double *rho) {
  PO_CHECK_PTR(ip);
  if (fobj) *fobj = ip->ip->fobj;
  if (rho) *rho = ip->ip->rho_penalty_search;
  return PO_OK;
}
int po_ip_set_penalty_gamma(po_ip ip, double gamma) {
  PO_CHECK_PTR(ip);
  ip->ip->setPenaltyGamma(gamma);
  return PO_OK;
}
int po_ip_set_penalty_gamma_array(po_ip ip, const double *gamma) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(gamma);
  ip->ip->setPenaltyGammaArray(gamma);
  return PO_OK;
}
int po_ip_set_quasi_newton(po_ip ip, po_qn qn) {
  PO_CHECK_PTR(ip);
  return ip->ip->setQuasiNewton(qn ? qn->qn : nullptr);
}
int po_ip_reset_problem_instance(po_ip ip, po_problem prob) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(prob);
  return ip->ip->resetProblemInstance(prob->p);
}
int po_ip_get_hvec_count(po_ip ip, int *nhvec) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(nhvec);
  *nhvec = ip->ip->nhvec;
  return PO_OK;
}
int po_ip_reset_design_and_bounds(po_ip ip) {
  PO_CHECK_PTR(ip);
  return ip->ip->resetDesignAndBounds();
}
int po_ip_reset_quasi_newton(po_ip ip) {
  PO_CHECK_PTR(ip);
  ip->ip->resetQuasiNewtonHessian();
  return PO_OK;
}
int po_ip_get_quasi_newton(po_ip ip, po_qn *qn) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(qn);
  *qn = ip->ip->qn ? &ip->ip->qn_handle : nullptr;
  return PO_OK;
}
int po_ip_write_solution_file(po_ip ip, const char *filename) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(filename);
  return ip->ip->writeSolutionFile(filename);
}
int po_ip_read_solution_file(po_ip ip, const char *filename) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(filename);
  return ip->ip->readSolutionFile(filename);
}
int po_ip_set_iteration_callback(po_ip ip, po_ip_iteration_fn fn, void *user) {
  PO_CHECK_PTR(ip);
  ip->ip->iter_cb = fn;
  ip->ip->iter_cb_user = user;
  return PO_OK;
}
int po_ip_get_history(po_ip ip, const char **text) {
  PO_CHECK_PTR(ip);
  *text = ip->ip->history.c_str();
  return PO_OK;
}
int po_ip_set_callback_timing(po_ip ip, int on) {
  PO_CHECK_PTR(ip);
  ip->ip->user_timing = on != 0;
  return PO_OK;
}
int po_ip_get_phase_times(po_ip ip, const char **names, const double **seconds, int *count) {
  PO_CHECK_PTR(ip);
  InteriorPoint *p = ip->ip;
  p->phase_names_joined.clear();
  for (size_t i = 0; i < p->phase_names.size(); i++) {
    if (i) p->phase_names_joined += ";";
    p->phase_names_joined += p->phase_names[i];
  }
  if (names) *names = p->phase_names_joined.c_str();
  if (seconds) *seconds = p->phase_seconds.data();
  if (count) *count = (int)p->phase_seconds.size();
  return PO_OK;
}
int po_ip_get_debug_ints(po_ip ip, const int **gpiv, int *ngpiv, int *check_flag, int64_t clamped[8]) {
  PO_CHECK_PTR(ip);
  InteriorPoint *p = ip->ip;
  if (gpiv) *gpiv = p->gPivots().data();
  if (ngpiv) *ngpiv = (int)p->gPivots().size();
  if (check_flag) *check_flag = p->checkFlag();
  if (clamped) {
    double out[8];
    PO_TRY(p->clampCounts(out));
    for (int i = 0; i < 8; i++) clamped[i] = (int64_t)(out[i] + 0.5);
  }
  return PO_OK;
}
int po_ip_get_bounds(po_ip ip, po_vec *lb, po_vec *ub) {
  PO_CHECK_PTR(ip);
  if (lb) *lb = static_cast<po_vec>(ip->ip->lowerBounds());
  if (ub) *ub = static_cast<po_vec>(ip->ip->upperBounds());
  return PO_OK;
}
int po_ip_debug_kkt_step(po_ip ip, double mu, po_vec *px, po_vec *pzl, po_vec *pzu,
                         const double **pz, const double **ps, const double **pt,
                         const double **pzs, const double **pzt) {
  PO_CHECK_PTR(ip);
  InteriorPoint *p = ip->ip;
  PO_TRY(p->debugKKTStep(mu));
  if (px) *px = static_cast<po_vec>(p->px);
  if (pzl) *pzl = static_cast<po_vec>(p->pzl);
  if (pzu) *pzu = static_cast<po_vec>(p->pzu);
  if (pz) *pz = p->step.z.data();
  if (ps) *ps = p->step.s.data();
  if (pt) *pt = p->step.t.data();
  if (pzs) *pzs = p->step.zs.data();
  if (pzt) *pzt = p->step.zt.data();
  return PO_OK;
}

int po_ip_debug_set_state(po_ip ip, const double *z, const double *s, const double *t, const double *zs,
                          const double *zt, double mu) {
  PO_CHECK_PTR(ip);
  InteriorPoint *p = ip->ip;
  if (p->c > 0) {
    PO_CHECK_PTR(z);
    PO_CHECK_PTR(s);
    PO_CHECK_PTR(t);
    PO_CHECK_PTR(zs);
    PO_CHECK_PTR(zt);
  }
  return p->debugSetState(z, s, t, zs, zt, mu);
}
int po_ip_debug_kkt(po_ip ip, double mu, int mode, double tau, po_ip_kkt_dump *out) {
  PO_CHECK_PTR(ip);
  PO_CHECK_PTR(out);
  InteriorPoint *p = ip->ip;
  if (mode < 0 || mode > 2) return PO_ERR_ARG;
  PO_TRY(p->debugKKT(mu, mode, tau));
  memset(out, 0, sizeof(*out));
  out->c = p->c;
  out->k = (int)p->cPivots().size();
  out->Dinv = static_cast<po_vec>(p->dinvVec());
  out->res_x = static_cast<po_vec>(p->rxVec());
  out->res_z = p->res.z.data();
  out->res_s = p->res.s.data();
  out->res_t = p->res.t.data();
  out->res_zs = p->res.zs.data();
  out->res_zt = p->res.zt.data();
  for (int i = 0; i < 4; i++) out->res_norms[i] = p->debug_norms[i];
  out->W = p->gramMatrix().data();
  out->G = p->Gmat0.data();
  out->Ce = p->Ce0.data();
  out->gpiv = p->gPivots().data();
  out->cpiv = p->cPivots().data();
  out->px = static_cast<po_vec>(p->px);
  out->pzl = static_cast<po_vec>(p->pzl);
  out->pzu = static_cast<po_vec>(p->pzu);
  out->pz = p->step.z.data();
  out->ps = p->step.s.data();
  out->pt = p->step.t.data();
  out->pzs = p->step.zs.data();
  out->pzt = p->step.zt.data();
  out->step_mins[0] = p->stepMins()[0];
  out->step_mins[1] = p->stepMins()[1];
  return PO_OK;
}

int po_ip_debug_kkt_step_sparse(po_ip ip, po_vec *pzw, po_vec *psw, po_vec *ptw, po_vec *pzsw, po_vec *pztw) {
  PO_CHECK_PTR(ip);
  InteriorPoint *p = ip->ip;
  po_vec *out[5] = {pzw, psw, ptw, pzsw, pztw};
  for (int i = 0; i < 5; i++) {
    if (out[i]) *out[i] = p->has_w ? static_cast<po_vec>(p->wstepv[i]) : nullptr;
  }
  return PO_OK;
}

// ---- standalone hot kernels -----------------------------------------------------------------------
static int gather_ptrs(po_vec ref, const po_vec *vecs, int nvecs, std::vector<const double *> &p) {
  p.resize(nvecs > 0 ? nvecs : 1);
  for (int j = 0; j < nvecs; j++) {
    PO_CHECK_PTR(vecs[j]);
    PO_TRY(same_layout(ref, vecs[j]));
    p[j] = vecs[j]->d;
  }
  return PO_OK;
}
int po_wgram(po_vec d, const po_vec *vecs, int nvecs, double *W) {
  PO_CHECK_PTR(d);
  PO_CHECK_PTR(W);
  std::vector<const double *> p;
  PO_TRY(gather_ptrs(d, vecs, nvecs, p));
  return k_wgram(d->ctx, d->d, p.data(), nvecs, d->n, W);
}
int po_wgram_with_rhs(po_vec d, const po_vec *vecs, int nvecs, double *W) {
  PO_CHECK_PTR(d);
  PO_CHECK_PTR(W);
  std::vector<const double *> p;
  PO_TRY(gather_ptrs(d, vecs, nvecs, p));
  return k_wgram(d->ctx, d->d, p.data(), nvecs, d->n, W, nullptr, nullptr, 0, 0.0, 1);
}
static int fail_arg(const char *msg) {
  set_error("%s", msg);
  return PO_ERR_ARG;
}
int po_group_panel(po_vec d, const po_vec *vecs, int ncols, int64_t nwcon, int nw, int skip, double alpha,
                   const po_vec *U) {
  PO_CHECK_PTR(d);
  std::vector<const double *> p;
  PO_TRY(gather_ptrs(d, vecs, ncols, p));
  std::vector<double *> u;
  for (int j = 0; j < ncols; j++) {
    if (!U || !U[j] || U[j]->n < nwcon) return fail_arg("po_group_panel: U vectors must hold nwcon entries");
    u.push_back(U[j]->d);
  }
  GroupMap m;
  m.nwcon = nwcon;
  m.start = 0;
  m.nw = nw;
  m.skip = skip;
  if (nwcon < 0 || nw <= 0 || skip < 0 || (nwcon > 0 && (nwcon - 1) * (int64_t)(nw + skip) + nw > d->n))
    return fail_arg("po_group_panel: the groups do not fit the vectors");
  return k_group_panel(d->ctx, m, p.data(), ncols, d->d, alpha, u.data());
}
int po_wgram_with_groups(po_vec d, const po_vec *vecs, int nvecs, int preweighted_last, int64_t nwcon, int nw,
                         int skip, double alpha, const po_vec *U, int ncols, double *W, int *fused) {
  PO_CHECK_PTR(d);
  PO_CHECK_PTR(W);
  PO_CHECK_PTR(fused);
  std::vector<const double *> p;
  PO_TRY(gather_ptrs(d, vecs, nvecs, p));
  std::vector<double *> u;
  for (int j = 0; j < ncols; j++) {
    if (!U || !U[j] || U[j]->n < nwcon) return fail_arg("po_wgram_with_groups: U vectors must hold nwcon entries");
    u.push_back(U[j]->d);
  }
  if (nwcon < 0 || nw <= 0 || skip < 0 || ncols > nvecs ||
      (nwcon > 0 && (nwcon - 1) * (int64_t)(nw + skip) + nw > d->n))
    return fail_arg("po_wgram_with_groups: the groups do not fit the vectors");
  GramGroups g;
  g.nwcon = nwcon;
  g.start = 0;
  g.nw = nw;
  g.skip = skip;
  g.alpha = alpha;
  g.ncols = ncols;
  g.U = u.data();
  bool done = false;
  PO_TRY(k_wgram(d->ctx, d->d, p.data(), nvecs, d->n, W, nullptr, nullptr, 0, 0.0, preweighted_last ? 1 : 0, false, &g,
                 &done));
  *fused = done ? 1 : 0;
  return PO_OK;
}
int po_bench_mdot(po_vec x, const po_vec *vecs, int nvecs, int reps, double *avg_ms, double *out) {
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(avg_ms);
  std::vector<const double *> p;
  PO_TRY(gather_ptrs(x, vecs, nvecs, p));
  Ctx *c = x->ctx;
  int grid = 0;
  PO_TRY(k_mdot_launch(c, x->d, p.data(), nvecs, x->n, &grid));  // warm-up, sizes partials
  PO_HIP(hipStreamSynchronize(c->stream));
  PO_HIP(hipEventRecord(c->ev0, c->stream));
  for (int r = 0; r < reps; r++) PO_TRY(k_mdot_launch(c, x->d, p.data(), nvecs, x->n, &grid));
  PO_HIP(hipEventRecord(c->ev1, c->stream));
  PO_HIP(hipEventSynchronize(c->ev1));
  float ms = 0.f;
  PO_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *avg_ms = (double)ms / (reps > 0 ? reps : 1);
  if (out) PO_TRY(reduce_finish(c, grid, nvecs, 0, 0, out, true));
  return PO_OK;
}
int po_bench_stream(po_vec x, po_vec y, int kind, int reps, double *avg_ms) {
  PO_CHECK_PTR(x);
  PO_CHECK_PTR(y);
  PO_CHECK_PTR(avg_ms);
  if (x->n != y->n || x->ctx != y->ctx) return PO_ERR_ARG;
  Ctx *c = x->ctx;
  PO_TRY(k_stream_launch(c, kind, x->d, y->d, x->n));  // warm-up
  PO_HIP(hipStreamSynchronize(c->stream));
  PO_HIP(hipEventRecord(c->ev0, c->stream));
  for (int r = 0; r < reps; r++) PO_TRY(k_stream_launch(c, kind, x->d, y->d, x->n));
  PO_HIP(hipEventRecord(c->ev1, c->stream));
  PO_HIP(hipEventSynchronize(c->ev1));
  float ms = 0.f;
  PO_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *avg_ms = (double)ms / (reps > 0 ? reps : 1);
  return PO_OK;
}
int po_bench_wgram(po_vec d, const po_vec *vecs, int nvecs, int reps, double *avg_ms) {
  PO_CHECK_PTR(d);
  PO_CHECK_PTR(avg_ms);
  std::vector<const double *> p;
  PO_TRY(gather_ptrs(d, vecs, nvecs, p));
  Ctx *c = d->ctx;
  int grid = 0, nslots = 0;
  PO_TRY(k_wgram_launch(c, d->d, p.data(), nvecs, d->n, &grid, &nslots));
  PO_HIP(hipStreamSynchronize(c->stream));
  PO_HIP(hipEventRecord(c->ev0, c->stream));
  for (int r = 0; r < reps; r++) PO_TRY(k_wgram_launch(c, d->d, p.data(), nvecs, d->n, &grid, &nslots));
  PO_HIP(hipEventRecord(c->ev1, c->stream));
  PO_HIP(hipEventSynchronize(c->ev1));
  float ms = 0.f;
  PO_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
  *avg_ms = (double)ms / (reps > 0 ? reps : 1);
  return PO_OK;
}

}  // extern "C"
