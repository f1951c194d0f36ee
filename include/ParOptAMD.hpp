// ParOptAMD.hpp -- header-only C++ facade over the C ABI (paropt_amd.h) with ParOpt's class and
// method names, so that problem classes written against the reference's C++ API
// (src/ParOptVec.h:28-98, src/ParOptProblem.h:42-296, src/ParOptQuasiNewton.h:32-220,
// src/ParOptOptions.h:9-61, src/ParOptInteriorPoint.h:128-217) port by changing the include and the
// communicator argument: the MPI_Comm of the reference becomes a po_ctx (one per GPU/rank).
//
// Semantics kept from the reference:
//   * intrusive reference counting: objects are born with count 0, holders incref(), decref()
//     deletes at 0 (ParOptBase, src/ParOptVec.h:28-47);
//   * ParOptVec::getArray returns the local length and a raw HOST pointer the caller may read and
//     write (src/ParOptVec.cpp:212-217).  Here that pointer is a pinned mirror of the HBM data: the
//     facade downloads a vector before handing it to a user callback as an input and uploads the
//     vectors a callback is documented to fill (x, lb, ub / g, Ac) when it returns;
//   * user callbacks return int fail (0 = ok); optimize() returns 0, 1 (mis-configuration) or the
//     initial evaluation's fail code; options are set by name with typed setOption overloads that
//     return non-zero for unknown names / wrong types (src/ParOptOptions.cpp:310-386).
// Only dense constraints (nwcon = 0) are supported this round.
#ifndef PAROPT_AMD_HPP
#define PAROPT_AMD_HPP

#include <stdio.h>

#include <map>
#include <string>
#include <vector>

extern "C" {
#include "paropt_amd.h"
}

typedef double ParOptScalar;

class ParOptBase {
 public:
  ParOptBase() : ref_count(0) {}
  virtual ~ParOptBase() {}
  void incref() { ref_count++; }
  void decref() {
    ref_count--;
    if (ref_count == 0) delete this;
  }

 private:
  int ref_count;
};

// ---- ParOptVec -----------------------------------------------------------------------------------
class ParOptVec : public ParOptBase {
 public:
  // a new zero-filled vector of n local components (ParOptBasicVec ctor)
  ParOptVec(po_ctx ctx, int n) : owner(true), h(NULL) { po_vec_create(ctx, n, &h); }
  // adapter over a vector owned by the library (callback arguments, solver state)
  explicit ParOptVec(po_vec borrowed) : owner(false), h(borrowed) {}
  ~ParOptVec() {
    if (owner && h) po_vec_decref(h);
  }
  void set(ParOptScalar alpha) { po_vec_set(h, alpha); }
  void zeroEntries() { po_vec_zero(h); }
  void copyValues(ParOptVec *vec) { po_vec_copy(h, vec->h); }
  double norm() { double v = 0; po_vec_norm(h, &v); return v; }
  double maxabs() { double v = 0; po_vec_maxabs(h, &v); return v; }
  double l1norm() { double v = 0; po_vec_l1norm(h, &v); return v; }
  ParOptScalar dot(ParOptVec *vec) { double v = 0; po_vec_dot(h, vec->h, &v); return v; }
  void mdot(ParOptVec **vecs, int nvecs, ParOptScalar *output) {
    std::vector<po_vec> hs(nvecs > 0 ? nvecs : 1);
    for (int i = 0; i < nvecs; i++) hs[i] = vecs[i]->h;
    po_vec_mdot(h, hs.data(), nvecs, output);
  }
  void scale(ParOptScalar alpha) { po_vec_scale(h, alpha); }
  void axpy(ParOptScalar alpha, ParOptVec *x) { po_vec_axpy(h, alpha, x->h); }
  int getArray(ParOptScalar **array) {
    int64_t n = 0;
    po_vec_size(h, &n);
    if (array) po_vec_get_array(h, array);
    return (int)n;
  }
  // explicit mirror control for code that writes through getArray outside of a problem callback
  void syncToDevice() { po_vec_sync_to_device(h); }
  void syncToHost() { po_vec_sync_to_host(h); }
  po_vec handle() { return h; }

 private:
  bool owner;
  po_vec h;
};

// ---- ParOptOptions --------------------------------------------------------------------------------
class ParOptOptions : public ParOptBase {
 public:
  int setOption(const char *name, const char *value) { s[name] = value ? value : ""; return 0; }
  int setOption(const char *name, int value) { i[name] = value; return 0; }
  int setOption(const char *name, double value) { f[name] = value; return 0; }
  // forwarded (and validated) when the solver is created
  int apply(po_ip ip) {
    int bad = 0;
    for (auto &kv : s) bad |= po_ip_set_option_str(ip, kv.first.c_str(), kv.second.c_str());
    for (auto &kv : i) bad |= po_ip_set_option_int(ip, kv.first.c_str(), kv.second);
    for (auto &kv : f) bad |= po_ip_set_option_float(ip, kv.first.c_str(), kv.second);
    return bad;
  }
  // the same, into a trust-region driver (one registry for both option sets); `algorithm` is
  // ParOptOptimizer's own switch and is not forwarded
  int apply(po_tr tr) {
    int bad = 0;
    for (auto &kv : s)
      if (kv.first != "algorithm") bad |= po_tr_set_option_str(tr, kv.first.c_str(), kv.second.c_str());
    for (auto &kv : i) bad |= po_tr_set_option_int(tr, kv.first.c_str(), kv.second);
    for (auto &kv : f) bad |= po_tr_set_option_float(tr, kv.first.c_str(), kv.second);
    return bad;
  }
  int apply(po_mma mma) {
    int bad = 0;
    for (auto &kv : s)
      if (kv.first != "algorithm") bad |= po_mma_set_option_str(mma, kv.first.c_str(), kv.second.c_str());
    for (auto &kv : i) bad |= po_mma_set_option_int(mma, kv.first.c_str(), kv.second);
    for (auto &kv : f) bad |= po_mma_set_option_float(mma, kv.first.c_str(), kv.second);
    return bad;
  }
  const char *getStringOption(const char *name, const char *def) {
    std::map<std::string, std::string>::iterator it = s.find(name);
    return it == s.end() ? def : it->second.c_str();
  }

 private:
  std::map<std::string, std::string> s;
  std::map<std::string, int> i;
  std::map<std::string, double> f;
};

// ---- ParOptProblem --------------------------------------------------------------------------------
class ParOptProblem : public ParOptBase {
 public:
  explicit ParOptProblem(po_ctx _ctx)
      : ctx(_ctx), nvars(0), ncon(0), nwcon(0), ninequality(-1), nwinequality(-1), nwblock(1), hprob(NULL) {}
  // the nwblock of the reference's `new ParOptQuasiDefBlockMat(this, nwblock)` in createQuasiDefMat(): with
  // nwblock > 1 addSparseInnerProduct fills packed upper nwblock x nwblock blocks (before the first handle())
  void setSparseBlockSize(int _nwblock) { nwblock = _nwblock; }
  virtual ~ParOptProblem() {
    if (hprob) po_problem_destroy(hprob);
  }
  po_ctx getContext() { return ctx; }
  void setProblemSizes(int _nvars, int _ncon, int _nwcon) {
    nvars = _nvars;
    ncon = _ncon;
    nwcon = _nwcon;
    if (ninequality < 0) ninequality = ncon;
    if (nwinequality < 0) nwinequality = nwcon;
  }
  void setNumInequalities(int _ninequality, int _nwinequality) {
    ninequality = _ninequality;
    nwinequality = _nwinequality;
  }
  void getProblemSizes(int *_nvars, int *_ncon, int *_nwcon) {
    if (_nvars) *_nvars = nvars;
    if (_ncon) *_ncon = ncon;
    if (_nwcon) *_nwcon = nwcon;
  }
  virtual ParOptVec *createDesignVec() { return new ParOptVec(ctx, nvars); }

  virtual void getVarsAndBounds(ParOptVec *x, ParOptVec *lb, ParOptVec *ub) = 0;
  virtual int evalObjCon(ParOptVec *x, ParOptScalar *fobj, ParOptScalar *cons) = 0;
  virtual int evalObjConGradient(ParOptVec *x, ParOptVec *g, ParOptVec **Ac) = 0;
  virtual void computeQuasiNewtonUpdateCorrection(ParOptVec *, ParOptScalar *, ParOptVec *, ParOptVec *,
                                                  ParOptVec *) {}
  virtual void writeOutput(int, ParOptVec *) {}
  // sparse constraints, nwblock = 1 (src/ParOptProblem.h:215-262); `out` / `pzw` are w-sized
  virtual void evalSparseCon(ParOptVec *, ParOptVec *) {}
  virtual void addSparseJacobian(ParOptScalar, ParOptVec *, ParOptVec *, ParOptVec *) {}
  virtual void addSparseJacobianTranspose(ParOptScalar, ParOptVec *, ParOptVec *, ParOptVec *) {}
  virtual void addSparseInnerProduct(ParOptScalar, ParOptVec *, ParOptVec *, ParOptScalar *) {}

  // the C-callback problem handed to the library (created on first use)
  po_problem handle() {
    if (!hprob) {
      po_problem_callbacks cb;
      cb.user = this;
      cb.get_vars_and_bounds = &ParOptProblem::tramp_vars;
      cb.eval_obj_con = &ParOptProblem::tramp_eval;
      cb.eval_obj_con_gradient = &ParOptProblem::tramp_grad;
      cb.qn_update_correction = NULL;
      cb.write_output = &ParOptProblem::tramp_write;
      if (po_problem_create_callbacks(ctx, nvars, ncon, ninequality, &cb, &hprob) != 0) {
        fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
      }
      if (hprob) attachSparse();
    }
    return hprob;
  }

 protected:
  po_ctx ctx;
  int nvars, ncon, nwcon, ninequality, nwinequality, nwblock;
  po_problem hprob;
  // registers the sparse-constraint callbacks with the library; ParOptSparseProblem registers its CSR form
  virtual void attachSparse() {
    if (nwcon <= 0) return;
    po_problem_sparse_callbacks scb;
    scb.eval_sparse_con = &ParOptProblem::tramp_wcon;
    scb.add_sparse_jacobian = &ParOptProblem::tramp_wjac;
    scb.add_sparse_jacobian_transpose = &ParOptProblem::tramp_wjact;
    scb.add_sparse_inner_product = &ParOptProblem::tramp_winner;
    if (po_problem_set_sparse_callbacks(hprob, nwcon, nwinequality, &scb) != 0 ||
        (nwblock > 1 && po_problem_set_sparse_block_size(hprob, nwblock) != 0)) {
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    }
  }

 private:
  static int tramp_vars(void *self, po_vec x, po_vec lb, po_vec ub) {
    ParOptVec vx(x), vl(lb), vu(ub);
    double *p;
    vx.getArray(&p);
    vl.getArray(&p);
    vu.getArray(&p);
    static_cast<ParOptProblem *>(self)->getVarsAndBounds(&vx, &vl, &vu);
    vx.syncToDevice();
    vl.syncToDevice();
    vu.syncToDevice();
    return 0;
  }
  static int tramp_eval(void *self, po_vec x, double *fobj, double *cons) {
    ParOptVec vx(x);
    vx.syncToHost();
    return static_cast<ParOptProblem *>(self)->evalObjCon(&vx, fobj, cons);
  }
  static int tramp_grad(void *self, po_vec x, po_vec g, const po_vec *Ac) {
    ParOptProblem *me = static_cast<ParOptProblem *>(self);
    ParOptVec vx(x), vg(g);
    vx.syncToHost();
    std::vector<ParOptVec *> va(me->ncon > 0 ? me->ncon : 1, (ParOptVec *)NULL);
    double *p;
    vg.getArray(&p);
    // Ac == NULL: the problem declared linear constraints (setLinearConstraints) and only g is wanted
    for (int j = 0; Ac && j < me->ncon; j++) {
      va[j] = new ParOptVec(Ac[j]);
      va[j]->incref();
      // the reference hands out zero-initialised Ac that the problem may fill sparsely
      va[j]->getArray(&p);
    }
    int fail = me->evalObjConGradient(&vx, &vg, Ac ? va.data() : NULL);
    vg.syncToDevice();
    for (int j = 0; Ac && j < me->ncon; j++) {
      va[j]->syncToDevice();
      va[j]->decref();
    }
    return fail;
  }
  static int tramp_wcon(void *self, po_vec x, po_vec out) {
    ParOptVec vx(x), vo(out);
    double *p;
    vx.syncToHost();
    vo.getArray(&p);
    static_cast<ParOptProblem *>(self)->evalSparseCon(&vx, &vo);
    vo.syncToDevice();
    return 0;
  }
  static int tramp_wjac(void *self, double alpha, po_vec x, po_vec px, po_vec out) {
    ParOptVec vx(x), vp(px), vo(out);
    vx.syncToHost();
    vp.syncToHost();
    vo.syncToHost();
    static_cast<ParOptProblem *>(self)->addSparseJacobian(alpha, &vx, &vp, &vo);
    vo.syncToDevice();
    return 0;
  }
  static int tramp_wjact(void *self, double alpha, po_vec x, po_vec pzw, po_vec out) {
    ParOptVec vx(x), vp(pzw), vo(out);
    vx.syncToHost();
    vp.syncToHost();
    vo.syncToHost();
    static_cast<ParOptProblem *>(self)->addSparseJacobianTranspose(alpha, &vx, &vp, &vo);
    vo.syncToDevice();
    return 0;
  }
  static int tramp_winner(void *self, double alpha, po_vec x, po_vec cvec, po_vec A) {
    ParOptVec vx(x), vc(cvec), va(A);
    vx.syncToHost();
    vc.syncToHost();
    va.syncToHost();
    double *a;
    va.getArray(&a);
    static_cast<ParOptProblem *>(self)->addSparseInnerProduct(alpha, &vx, &vc, a);
    va.syncToDevice();
    return 0;
  }
  static int tramp_write(void *self, int iter, po_vec x) {
    ParOptVec vx(x);
    static_cast<ParOptProblem *>(self)->writeOutput(iter, &vx);
    return 0;
  }
};

// ---- ParOptSparseProblem (src/ParOptProblem.h:301-395): fixed CSR pattern for the sparse Jacobian -------------
// Overlapping rows are allowed; the quasi-definite system is solved with the device sparse Cholesky
// (ParOptQuasiDefSparseMat of the reference).  `data` is a HOST array of nnz entries in the order of cols, as in
// the reference; it is uploaded after the call.
class ParOptSparseProblem : public ParOptProblem {
 public:
  explicit ParOptSparseProblem(po_ctx _ctx) : ParOptProblem(_ctx) {}
  // after setProblemSizes(), as in the reference (:306-312)
  void setSparseJacobianData(const int *_rowp, const int *_cols) {
    rowp.assign(_rowp, _rowp + nwcon + 1);
    cols.assign(_cols, _cols + rowp[nwcon]);
    data.assign(cols.size() > 0 ? cols.size() : 1, 0.0);
  }
  int getSparseJacobianData(const int **_rowp, const int **_cols, const ParOptScalar **_data) {
    if (_rowp) *_rowp = rowp.data();
    if (_cols) *_cols = cols.data();
    if (_data) *_data = data.data();
    return (int)cols.size();
  }
  virtual int evalSparseObjCon(ParOptVec *x, ParOptScalar *fobj, ParOptScalar *cons, ParOptVec *sparse_con) = 0;
  virtual int evalSparseObjConGradient(ParOptVec *x, ParOptVec *g, ParOptVec **Ac, ParOptScalar *data) = 0;
  // the base-class evaluations are never reached: the library calls the sparse forms (:348-358)
  int evalObjCon(ParOptVec *, ParOptScalar *, ParOptScalar *) { return 1; }
  int evalObjConGradient(ParOptVec *, ParOptVec *, ParOptVec **) { return 1; }
  // one line about the device factorization (ParOptQuasiDefMat::getFactorInfo)
  const char *getFactorInfo() { return hprob ? po_quasidef_factor_info(hprob) : NULL; }

 protected:
  void attachSparse() {
    if (po_problem_set_sparse_jacobian_data(hprob, nwcon, nwinequality, rowp.data(), cols.data(),
                                            &ParOptSparseProblem::tramp_sobjcon,
                                            &ParOptSparseProblem::tramp_sgrad) != 0) {
      fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    }
  }

 private:
  std::vector<int> rowp, cols;
  std::vector<ParOptScalar> data;
  static int tramp_sobjcon(void *self, po_vec x, double *fobj, double *cons, po_vec sparse) {
    ParOptVec vx(x), vs(sparse);
    double *p;
    vx.syncToHost();
    vs.getArray(&p);
    int fail = static_cast<ParOptSparseProblem *>(self)->evalSparseObjCon(&vx, fobj, cons, &vs);
    vs.syncToDevice();
    return fail;
  }
  static int tramp_sgrad(void *self, po_vec x, po_vec g, const po_vec *Ac, double *ddata, int64_t nnz) {
    ParOptSparseProblem *me = static_cast<ParOptSparseProblem *>(self);
    ParOptVec vx(x), vg(g);
    vx.syncToHost();
    std::vector<ParOptVec *> va(me->ncon > 0 ? me->ncon : 1, (ParOptVec *)NULL);
    double *p;
    vg.getArray(&p);
    for (int j = 0; j < me->ncon; j++) {
      va[j] = new ParOptVec(Ac[j]);
      va[j]->incref();
      va[j]->getArray(&p);
    }
    int fail = me->evalSparseObjConGradient(&vx, &vg, va.data(), me->data.data());
    vg.syncToDevice();
    for (int j = 0; j < me->ncon; j++) {
      va[j]->syncToDevice();
      va[j]->decref();
    }
    if (po_ctx_memcpy(me->ctx, ddata, me->data.data(), (int64_t)sizeof(double) * nnz, 1) != 0) return 1;
    return fail;
  }
};

// ---- compact quasi-Newton -----------------------------------------------------------------------
enum ParOptBFGSUpdateType { PAROPT_SKIP_NEGATIVE_CURVATURE, PAROPT_DAMPED_UPDATE };
enum ParOptQuasiNewtonDiagonalType {
  PAROPT_YTY_OVER_YTS,
  PAROPT_YTS_OVER_STS,
  PAROPT_INNER_PRODUCT_YTY_OVER_YTS,
  PAROPT_INNER_PRODUCT_YTS_OVER_STS
};

class ParOptCompactQuasiNewton : public ParOptBase {
 public:
  ~ParOptCompactQuasiNewton() {
    for (ParOptVec *v : zwrap) v->decref();
    if (h) po_qn_destroy(h);
  }
  void setInitDiagonalType(ParOptQuasiNewtonDiagonalType t) {
    po_qn_set_diag_type(h, t == PAROPT_YTS_OVER_STS ? PO_QN_YTS_OVER_STS : PO_QN_YTY_OVER_YTS);
  }
  void reset() { po_qn_reset(h); }
  int update(ParOptVec *, const ParOptScalar *, ParOptVec *, ParOptVec *s, ParOptVec *y) {
    int rc = 0;
    po_qn_update(h, s->handle(), y->handle(), &rc);
    return rc;
  }
  void mult(ParOptVec *x, ParOptVec *y) { po_qn_mult(h, x->handle(), y->handle()); }
  void multAdd(ParOptScalar alpha, ParOptVec *x, ParOptVec *y) { po_qn_mult_add(h, alpha, x->handle(), y->handle()); }
  int getCompactMat(ParOptScalar *b0, const ParOptScalar **d, const ParOptScalar **M, ParOptVec ***Z) {
    int k = 0;
    const po_vec *zs = NULL;
    po_qn_get_compact(h, &k, b0, d, M, &zs);
    if (Z) {
      for (ParOptVec *v : zwrap) v->decref();
      zwrap.clear();
      for (int i = 0; i < k; i++) {
        zwrap.push_back(new ParOptVec(zs[i]));
        zwrap.back()->incref();
      }
      *Z = zwrap.data();
    }
    return k;
  }
  int getMaxLimitedMemorySize() { int k = 0; po_qn_max_size(h, &k); return k; }
  po_qn handle() { return h; }

 protected:
  ParOptCompactQuasiNewton() : h(NULL) {}
  po_qn h;
  std::vector<ParOptVec *> zwrap;
};

class ParOptLBFGS : public ParOptCompactQuasiNewton {
 public:
  ParOptLBFGS(ParOptProblem *prob, int subspace) {
    int n;
    prob->getProblemSizes(&n, NULL, NULL);
    po_qn_create(prob->getContext(), PO_QN_BFGS, n, subspace, &h);
  }
  void setBFGSUpdateType(ParOptBFGSUpdateType t) {
    po_qn_set_update_type(h, t == PAROPT_DAMPED_UPDATE ? PO_BFGS_DAMPED_UPDATE : PO_BFGS_SKIP_NEGATIVE_CURVATURE);
  }
};

class ParOptLSR1 : public ParOptCompactQuasiNewton {
 public:
  ParOptLSR1(ParOptProblem *prob, int subspace) {
    int n;
    prob->getProblemSizes(&n, NULL, NULL);
    po_qn_create(prob->getContext(), PO_QN_SR1, n, subspace, &h);
  }
};

// ---- ParOptInteriorPoint ------------------------------------------------------------------------
class ParOptInteriorPoint : public ParOptBase {
 public:
  ParOptInteriorPoint(ParOptProblem *_prob, ParOptOptions *_options = NULL)
      : prob(_prob), options(_options), ip(NULL), x(NULL), zl(NULL), zu(NULL), zw(NULL), sw(NULL), tw(NULL) {
    prob->incref();
    if (options) options->incref();
    if (po_ip_create(prob->handle(), &ip) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    if (ip && options && options->apply(ip) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
  }
  ~ParOptInteriorPoint() {
    if (x) x->decref();
    if (zl) zl->decref();
    if (zu) zu->decref();
    if (zw) zw->decref();
    if (sw) sw->decref();
    if (tw) tw->decref();
    if (ip) po_ip_destroy(ip);
    if (options) options->decref();
    prob->decref();
  }
  ParOptProblem *getOptProblem() { return prob; }
  int optimize(const char *checkpoint = NULL) { return ip ? po_ip_optimize(ip, checkpoint) : 1; }
  void getProblemSizes(int *nvars, int *ncon, int *nwcon) { prob->getProblemSizes(nvars, ncon, nwcon); }
  // borrowed internals, as in the reference (src/ParOptInteriorPoint.cpp:793-826)
  void getOptimizedPoint(ParOptVec **_x, ParOptScalar **_z, ParOptVec **_zw, ParOptVec **_zl, ParOptVec **_zu) {
    po_vec hx, hzl, hzu;
    const double *z;
    po_ip_get_optimized_point(ip, &hx, &z, &hzl, &hzu);
    wrap(&x, hx);
    wrap(&zl, hzl);
    wrap(&zu, hzu);
    if (_x) *_x = x;
    if (_z) *_z = const_cast<double *>(z);
    if (_zw) {
      po_vec hzw = NULL;
      po_ip_get_optimized_sparse(ip, &hzw, NULL, NULL, NULL, NULL);
      wrap(&zw, hzw);
      *_zw = zw;
    }
    if (_zl) *_zl = zl;
    if (_zu) *_zu = zu;
  }
  void getOptimizedSlacks(ParOptScalar **s, ParOptScalar **t, ParOptVec **_sw, ParOptVec **_tw) {
    const double *ps, *pt, *pzs, *pzt;
    po_ip_get_optimized_slacks(ip, &ps, &pt, &pzs, &pzt);
    if (s) *s = const_cast<double *>(ps);
    if (t) *t = const_cast<double *>(pt);
    if (_sw || _tw) {
      po_vec hsw = NULL, htw = NULL;
      po_ip_get_optimized_sparse(ip, NULL, &hsw, &htw, NULL, NULL);
      wrap(&sw, hsw);
      wrap(&tw, htw);
      if (_sw) *_sw = sw;
      if (_tw) *_tw = tw;
    }
  }
  void getIterationCounters(int *niter = NULL, int *neval = NULL, int *ngeval = NULL, int *nhvec = NULL) {
    po_ip_get_counters(ip, niter, neval, ngeval);
    if (nhvec) po_ip_get_hvec_count(ip, nhvec);
  }
  double getBarrierParameter() { double v = 0; po_ip_get_barrier_parameter(ip, &v); return v; }
  ParOptScalar getComplementarity() { double v = 0; po_ip_get_complementarity(ip, &v); return v; }
  void setPenaltyGamma(double gamma) { po_ip_set_penalty_gamma(ip, gamma); }
  void setPenaltyGamma(const double *gamma) { po_ip_set_penalty_gamma_array(ip, gamma); }
  // the caller keeps ownership of (and must keep alive) the approximation; NULL detaches it
  void setQuasiNewton(ParOptCompactQuasiNewton *qn) { po_ip_set_quasi_newton(ip, qn ? qn->handle() : NULL); }
  void resetProblemInstance(ParOptProblem *problem) {
    if (po_ip_reset_problem_instance(ip, problem->handle()) == 0) {
      problem->incref();
      prob->decref();
      prob = problem;
    } else {
      fprintf(stderr, "ParOpt: Incompatible problem instance\n");
    }
  }
  void resetQuasiNewtonHessian() { po_ip_reset_quasi_newton(ip); }
  void resetDesignAndBounds() { po_ip_reset_design_and_bounds(ip); }
  int writeSolutionFile(const char *filename) { return po_ip_write_solution_file(ip, filename); }
  int readSolutionFile(const char *filename) { return po_ip_read_solution_file(ip, filename); }
  const char *getHistory() { const char *t = ""; po_ip_get_history(ip, &t); return t; }

 private:
  void wrap(ParOptVec **slot, po_vec h) {
    if (*slot) (*slot)->decref();
    *slot = NULL;
    if (h) {
      *slot = new ParOptVec(h);
      (*slot)->incref();
    }
  }
  ParOptProblem *prob;
  ParOptOptions *options;
  po_ip ip;
  ParOptVec *x, *zl, *zu, *zw, *sw, *tw;
};

// ---- ParOptOptimizer: algorithm = "ip" | "tr" | "mma" (src/ParOptOptimizer.cpp:65-206) --------------
// The reference's generic entry point.  "tr" builds the quasi-Newton object, the quadratic
// subproblem, the interior-point sub-solver and the trust-region driver exactly as :108-183 does.
class ParOptOptimizer : public ParOptBase {
 public:
  ParOptOptimizer(ParOptProblem *_prob, ParOptOptions *_options)
      : prob(_prob), options(_options), ip(NULL), tr(NULL), mma(NULL), x(NULL) {
    prob->incref();
    options->incref();
  }
  ~ParOptOptimizer() {
    if (x) x->decref();
    if (ip) ip->decref();
    if (tr) po_tr_destroy(tr);
    if (mma) po_mma_destroy(mma);
    options->decref();
    prob->decref();
  }
  void optimize() {
    const std::string algorithm = options->getStringOption("algorithm", "tr");
    if (algorithm == "ip") {
      if (!ip) {
        ParOptOptions *o = options;
        ip = new ParOptInteriorPoint(prob, o);
        ip->incref();
      }
      ip->optimize();
    } else if (algorithm == "tr") {
      if (!tr) {
        if (po_tr_create(prob->handle(), &tr) != 0 || options->apply(tr) != 0) {
          fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
          return;
        }
      }
      if (po_tr_optimize(tr) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    } else if (algorithm == "mma") {
      if (!mma) {
        if (po_mma_create(prob->handle(), &mma) != 0 || options->apply(mma) != 0) {
          fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
          return;
        }
      }
      if (po_mma_optimize(mma) != 0) fprintf(stderr, "ParOptAMD: %s\n", po_last_error());
    } else {
      fprintf(stderr, "ParOptOptimizer Error: Unrecognized algorithm option %s\n", algorithm.c_str());
    }
  }
  void getOptimizedPoint(ParOptVec **_x, ParOptScalar **_z, ParOptVec **_zw, ParOptVec **_zl,
                         ParOptVec **_zu) {
    if (tr || mma) {
      po_vec hx = NULL;
      const double *z = NULL;
      if (tr) {
        po_tr_get_optimized_point(tr, &hx, &z, NULL);
      } else {
        po_mma_get_optimized_point(mma, &hx, &z, NULL, NULL, NULL);
      }
      if (x) x->decref();
      x = new ParOptVec(hx);
      x->incref();
      if (_x) *_x = x;
      if (_z) *_z = const_cast<double *>(z);
      if (_zw) *_zw = NULL;
      if (_zl) *_zl = NULL;
      if (_zu) *_zu = NULL;
    } else if (ip) {
      ip->getOptimizedPoint(_x, _z, _zw, _zl, _zu);
    }
  }
  const char *getTrustRegionHistory() {
    const char *t = "";
    if (tr) po_tr_get_history(tr, &t);
    return t;
  }

 private:
  ParOptProblem *prob;
  ParOptOptions *options;
  ParOptInteriorPoint *ip;
  po_tr tr;
  po_mma mma;
  ParOptVec *x;
};

#endif  // PAROPT_AMD_HPP
