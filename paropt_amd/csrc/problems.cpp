// Problem implementations: the C-callback problem and the device-resident separable workloads.
#include <math.h>
#include <stdio.h>

#include <string>

#include "problem.hpp"

namespace po {

static void shard(int64_t n, int rank, int size, int64_t *nlocal, int64_t *offset) {
  const int64_t base = n / size, rem = n % size;
  *nlocal = base + (rank < rem ? 1 : 0);
  *offset = rank * base + (rank < rem ? rank : rem);
}

// ---- CSR form (ParOptSparseProblem) and the generic fallbacks ------------------------------------
int comm_allreduce_host(Ctx *c, double *values, int count, int op);  // context.cpp

// Rows of equal length nw with consecutive columns at equal spacing: the grouped pattern of problem.hpp
static bool recognise_groups(int64_t n, int64_t w, const int *rowp, const int *cols, GroupMap *gm) {
  if (w <= 0 || rowp[0] != 0) return false;
  const int64_t nw = rowp[1] - rowp[0];
  if (nw <= 0 || nw > (1 << 20)) return false;
  const int64_t start = cols[0];
  int64_t period = nw;
  if (w > 1) {
    if (rowp[2] - rowp[1] != nw) return false;
    period = (int64_t)cols[rowp[1]] - start;
  }
  if (start < 0 || period < nw || start + (w - 1) * period + nw > n) return false;
  for (int64_t i = 0; i < w; i++) {
    if (rowp[i + 1] - rowp[i] != nw) return false;
    const int *cr = cols + rowp[i];
    const int64_t c0 = start + i * period;
    for (int64_t k = 0; k < nw; k++)
      if (cr[k] != c0 + k) return false;
  }
  gm->nwcon = w;
  gm->start = start;
  gm->nw = (int)nw;
  gm->skip = (int)(period - nw);
  return true;
}

int Problem::setSparseJacobianData(int64_t nwcon_, int64_t nwineq_, const int *rowp, const int *cols) {
  if (nwcon_ < 0 || nwineq_ < 0 || nwineq_ > nwcon_ || !rowp || (rowp[nwcon_] > 0 && !cols)) {
    set_error("setSparseJacobianData: bad arguments");
    return PO_ERR_ARG;
  }
  static const bool no_groups = getenv("PAROPT_AMD_NO_CSR_GROUPS") != nullptr;
  GroupMap gm;
  bool rec = !no_groups && nwblock == 1 && ((nwcon_ == 0 && ctx->size > 1) || recognise_groups(nlocal, nwcon_, rowp, cols, &gm));
  if (ctx->size > 1) {  // all ranks or none (the two paths issue different reductions): collective
    double flag = rec ? 1.0 : 0.0;
    PO_TRY(comm_allreduce_host(ctx, &flag, 1, 1));
    rec = flag != 0.0;
  }
  CsrSparse *m = new CsrSparse(ctx, nlocal, nwcon_);
  int rc = rec ? m->setPatternLight(rowp, cols) : m->setPattern(rowp, cols);
  if (rc != PO_OK) {
    delete m;
    return rc;
  }
  delete csr;
  csr = m;
  nwcon = nwcon_;
  nwinequality = nwineq_;
  grouped = rec;
  gmap = rec ? gm : GroupMap();
  group_alpha = 0.0;  // known after the first gradient evaluation (csrValuesChanged)
  return PO_OK;
}

int Problem::csrValuesChanged() {
  if (!csr) return PO_OK;
  if (!grouped) return csr->valuesChanged();
  // every entry the same value?  {min, max} over the nnz values the user just wrote (one pass + host result)
  double mm[2] = {0.0, 0.0};
  PO_TRY(k_minmax(ctx, csr->data, csr->nnz, mm));  // over all ranks: every rank takes the same decision
  if (mm[0] > mm[1]) {  // no entries anywhere
    group_alpha = 0.0;
    return PO_OK;
  }
  if (mm[0] == mm[1]) {
    group_alpha = mm[0];
    return PO_OK;
  }
  // not the uniform weighting pattern after all: the general path from here on (analysis of the pattern once, device
  // sparse Cholesky), with the values already written
  if (ctx->rank == 0 && csr_group_fallbacks == 0)
    fprintf(stderr, "paropt_amd: the sparse Jacobian has the grouped pattern but entries in [%g, %g]: general CSR path\n",
            mm[0], mm[1]);
  csr_group_fallbacks++;
  grouped = false;
  gmap = GroupMap();
  PO_TRY(csr->upgradeFromLight());
  return csr->valuesChanged();
}
Problem::~Problem() {
  delete csr;
  vec_decref(blk);
  vec_decref(wones);
  if (blk_flag) (void)hipFree(blk_flag);
}
const char *Problem::sparseFactorInfo() {
  if (csr && !grouped) return csr->factorInfo();
  factor_info = "nblock: " + std::to_string(nwblock);  // ParOptQuasiDefBlockMat::getFactorInfo (:218-221)
  return factor_info.c_str();
}
int Problem::setSparseBlockSize(int nwblock_) {
  if (nwblock_ < 1 || nwblock_ > 16 || (nwcon % nwblock_) != 0 || csr) {
    set_error("sparse block size %d: must be 1..16, divide the %lld local sparse constraints, and the problem must "
              "not be in the CSR form", nwblock_, (long long)nwcon);
    return PO_ERR_ARG;
  }
  // allocate first, commit on success: a failure leaves the problem in its previous (consistent) state
  Vec *nblk = nullptr, *nones = nullptr;
  if (nwblock_ > 1) {
    nblk = vec_new(ctx, nwcon * (nwblock_ + 1) / 2);
    nones = vec_new(ctx, nwcon);
    int rc = (nblk && nones) ? k_fill(ctx, nones->d, nwcon, 1.0) : PO_ERR_HIP;
    if (rc == PO_OK && !blk_flag && hipMalloc((void **)&blk_flag, 4 * sizeof(int)) != hipSuccess) rc = PO_ERR_HIP;
    if (rc != PO_OK) {
      vec_decref(nblk);
      vec_decref(nones);
      return rc;
    }
  }
  vec_decref(blk);
  vec_decref(wones);
  blk = nblk;
  wones = nones;
  nwblock = nwblock_;
  blk_nwcon = nwcon;
  return PO_OK;
}
// ParOptSparseProblem::evalSparseCon copies the values the last evaluation stored (.cpp:750-760)
int Problem::evalSparseCon(Vec *, Vec *out) {
  if (!csr) return 0;
  return k_copy(ctx, out->d, csr->cw->d, nwcon) != PO_OK;
}
int Problem::addSparseJacobian(double alpha, Vec *, Vec *px, Vec *out) {  // .cpp:762-788
  if (grouped) return k_group_sum(ctx, gmap, out->d, 1, 0.0, alpha * group_alpha, px->d);
  if (!csr) return 0;
  return csr->spmv(alpha, px->d, out->d) != PO_OK;
}
int Problem::setSparseJacobianTranspose(double alpha, Vec *x, Vec *pzw, Vec *out) {
  if (grouped) return k_group_scatter_set(ctx, gmap, out->d, alpha * group_alpha, pzw->d, nlocal);
  PO_TRY(k_fill(ctx, out->d, out->n, 0.0));
  return addSparseJacobianTranspose(alpha, x, pzw, out) != 0 ? PO_ERR_USER : PO_OK;
}
int Problem::addSparseJacobianTranspose(double alpha, Vec *, Vec *pzw, Vec *out) {  // .cpp:790-816
  if (grouped) return k_group_scatter(ctx, gmap, out->d, alpha * group_alpha, pzw->d, nlocal);
  if (!csr) return 0;
  return csr->spmvT(alpha, pzw->d, out->d) != PO_OK;
}
int Problem::setSparseJacobian(double alpha, Vec *x, Vec *px, Vec *out) {
  // out = alpha Aw px = alpha group_alpha (group sums of px), every entry written
  if (grouped) return k_group_sum(ctx, gmap, out->d, 0, 0.0, alpha * group_alpha, px->d);
  PO_TRY(k_fill(ctx, out->d, nwcon, 0.0));
  return addSparseJacobian(alpha, x, px, out) != 0 ? PO_ERR_USER : PO_OK;
}
int Problem::sparseFactorFromSlacks(Vec *x, Vec *d, const WVars &v, Vec *cw) {
  // grouped with entries +-1: Cdiag from the slack blocks, the group sums of d and the reciprocal in ONE launch
  if (grouped && nwblock == 1 && group_alpha * group_alpha == 1.0) return k_group_factor(ctx, gmap, v, d->d, cw->d);
  PO_TRY(k_w_cdiag(ctx, v, nwcon, cw->d));
  return sparseFactor(x, d, cw);
}
int Problem::addSparseInnerProduct(double alpha, Vec *, Vec *cvec, Vec *A) {
  if (grouped) return k_group_sum(ctx, gmap, A->d, 1, 0.0, alpha * (group_alpha * group_alpha), cvec->d);
  if (!csr) return 0;
  return csr->innerProduct(alpha, cvec->d, A->d) != PO_OK;
}
int Problem::sparseFactor(Vec *x, Vec *d, Vec *cw) {
  // grouped: Cw = 1 / (Cdiag + alpha^2 (sum of d over the group)): the group sum and the reciprocal in one launch
  if (grouped && nwblock == 1) return k_group_sum(ctx, gmap, cw->d, 1, 0.0, group_alpha * group_alpha, d->d, 1);
  if (csr) return csr->factor(d->d, cw->d);
  if (nwblock > 1) {  // :60-112
    const int64_t nb = nwcon / nwblock;
    PO_TRY(k_blk_init(ctx, cw->d, nb, nwblock, blk->d));
    if (addSparseInnerProduct(1.0, x, d, blk) != 0) return PO_ERR_USER;
    const int flag0[4] = {0, 0x7fffffff, 0, 0};
    PO_HIP(hipMemcpyAsync(blk_flag, flag0, sizeof(flag0), hipMemcpyHostToDevice, ctx->stream));
    PO_TRY(k_blk_factor(ctx, blk->d, nb, nwblock, blk_flag));
    int flag[4] = {0, 0, 0, 0};
    PO_HIP(hipMemcpyAsync(flag, blk_flag, sizeof(flag), hipMemcpyDeviceToHost, ctx->stream));
    PO_HIP(hipStreamSynchronize(ctx->stream));
    if (flag[0] != 0) {  // the reference returns dpptrf's info, which its caller ignores (:1930)
      if (blk_breakdowns == 0 && ctx->rank == 0) {  // warn once, like the CSR form
        fprintf(stderr,
                "ParOpt warning: the sparse constraint matrix is not positive definite (first failing row %d, block "
                "%d; %d pivots replaced)\n", flag[1], flag[1] / nwblock, flag[0]);
      }
      blk_breakdowns++;
    }
    return PO_OK;
  }
  if (addSparseInnerProduct(1.0, x, d, cw) != 0) return PO_ERR_USER;
  return k_recip(ctx, cw->d, nwcon);
}
int Problem::sparseHalfSolve(double *const *U, int nv, Vec *cw, const double **weights) {
  if (csr && !grouped) {
    *weights = csr->unitWeights();
    return csr->halfSolve(U, nv);
  }
  if (nwblock > 1) {  // S = U^T U per block: U_panel^T S^-1 U_panel = (U^-T P)^T (U^-T P)
    *weights = wones->d;
    return k_blk_solve(ctx, blk->d, nwcon / nwblock, nwblock, U, nv, 1);
  }
  *weights = cw->d;
  return PO_OK;
}

int Problem::sparseCorrection(const double *const *U, int nv, const double *alpha, Vec *cw, Vec *out, Vec *acc) {
  if ((!csr || grouped) && nwblock == 1 && nv <= kMaxPanel) {
    // scalar block form: sum, scale by -cw and the caller's accumulation in ONE w-sized launch (round 4; the same
    // operations in the same order as the three launches below)
    return k_w_correction(ctx, U, nv, alpha, cw->d, nwcon, out->d, acc ? acc->d : nullptr);
  }
  if (csr && !grouped) {
    PO_TRY(csr->correction(U, nv, alpha, out->d));
  } else {
    PO_TRY(k_panel_axpy(ctx, out->d, 0.0, nullptr, 0.0, alpha, U, nv, nwcon));
    if (nwblock > 1) {  // -U^-1 (Y alpha)
      double *Y[1] = {out->d};
      PO_TRY(k_blk_solve(ctx, blk->d, nwcon / nwblock, nwblock, Y, 1, 2));
      PO_TRY(k_scale(ctx, out->d, nwcon, -1.0));
    } else {
      PO_TRY(k_mul(ctx, out->d, -1.0, cw->d, out->d, nwcon));
    }
  }
  if (acc) PO_TRY(k_axpy(ctx, acc->d, 1.0, out->d, nwcon));
  return PO_OK;
}

int Problem::sparseJacobianPanel(Vec *x, Vec *d, const double *const *P, int nv, double *const *U,
                                 Vec *work) {
  if (grouped) return k_group_panel(ctx, gmap, P, nv, d->d, group_alpha, U);
  if (csr) return csr->panelPermuted(d->d, P, nv, U);
  for (int j = 0; j < nv; j++) {
    PO_TRY(k_mul(ctx, work->d, 1.0, d->d, P[j], nlocal));
    PO_TRY(k_fill(ctx, U[j], nwcon, 0.0));
    // a real po_vec handle that borrows the panel column: host-side callbacks may ask it for a pinned mirror
    // (po_vec_get_array), which is released here -- the handle never reaches vec_decref
    po_vec_s u;
    u.ctx = ctx;
    u.n = nwcon;
    u.d = U[j];
    u.ref = 1;
    u.h = nullptr;
    const int fail = addSparseJacobian(1.0, x, work, &u);
    if (u.h) {
      (void)hipStreamSynchronize(ctx->stream);
      (void)hipHostFree(u.h);
      mirror_freed();
    }
    if (fail != 0) return PO_ERR_USER;
  }
  return PO_OK;
}

int Problem::checkGradients(double dh, Vec *x, bool check_hvec, Vec *xt, Vec *px, std::string *report) {
  const int64_t n = nlocal;
  Vec *g = vec_new(ctx, n), *gt = vec_new(ctx, n);
  std::vector<Vec *> A, At;
  bool ok = g && gt;
  for (int j = 0; ok && j < ncon; j++) {
    A.push_back(vec_new(ctx, n));
    At.push_back(vec_new(ctx, n));
    ok = A.back() && At.back();
  }
  int rc = ok ? PO_OK : PO_ERR_HIP;
  auto run = [&]() -> int {
    double f0 = 0.0, f1 = 0.0;
    std::vector<double> c0(ncon > 0 ? ncon : 1, 0.0), c1(c0), Ap(c0);
    if (evalObjCon(x, &f0, c0.data()) != 0 || evalObjConGradient(x, g, A.data()) != 0) return PO_ERR_USER;
    PO_TRY(k_sign(ctx, px->d, g->d, n));  // p_i = +1 where g_i >= 0, else -1
    double gp = 0.0;
    PO_TRY(k_reduce1(ctx, RED_DOT, g->d, px->d, n, &gp));
    std::vector<const double *> ap;
    for (Vec *v : A) ap.push_back(v->d);
    if (ncon > 0) PO_TRY(k_mdot(ctx, px->d, ap.data(), ncon, n, Ap.data()));
    const double one[1] = {dh};
    const double *pv[1] = {px->d};
    PO_TRY(k_panel_axpy(ctx, xt->d, 1.0, x->d, 0.0, one, pv, 1, n));
    if (evalObjCon(xt, &f1, c1.data()) != 0) return PO_ERR_USER;
    char line[256];
    auto row = [&](const char *what, int idx, double actual, double fd) {
      const double err = fabs(actual - fd), rel = err / (fabs(actual) > 1e-300 ? fabs(actual) : 1.0);
      if (idx < 0) {
        snprintf(line, sizeof(line), "%s\n%15s %15s %15s %15s\n%15.6e %15.6e %15.4e %15.4e\n", what, "Actual", "FD",
                 "Err", "Rel err", actual, fd, err, rel);
      } else {
        snprintf(line, sizeof(line), "%s[%d]\n%15.6e %15.6e %15.4e %15.4e\n", what, idx, actual, fd, err, rel);
      }
      *report += line;
    };
    *report += "\nGradient check (forward differences along sign(g))\n";
    row("Objective gradient test", -1, gp, (f1 - f0) / dh);
    for (int j = 0; j < ncon; j++) row("Constraint gradient test", j, Ap[j], (c1[j] - c0[j]) / dh);
    if (check_hvec) {
      std::vector<double> z(ncon > 0 ? ncon : 1, 0.0), mz(z);
      for (int j = 0; j < ncon; j++) {
        z[j] = 2.3 - 0.15 * (j % 5);
        mz[j] = -z[j];
      }
      if (evalObjConGradient(xt, gt, At.data()) != 0) return PO_ERR_USER;
      std::vector<const double *> atp;
      for (Vec *v : At) atp.push_back(v->d);
      // Lagrangian gradients g - sum z_j A_j (the sparse constraints of the built-in problems are linear or
      // enter through zw = 0 here)
      PO_TRY(k_panel_axpy(ctx, g->d, 0.0, nullptr, 1.0, mz.data(), ap.data(), ncon, n));
      PO_TRY(k_panel_axpy(ctx, gt->d, 0.0, nullptr, 1.0, mz.data(), atp.data(), ncon, n));
      if (evalHvecProduct(x, z.data(), nullptr, px, xt) == 0) {
        PO_TRY(k_axpy(ctx, gt->d, -1.0, g->d, n));
        PO_TRY(k_scale(ctx, gt->d, n, 1.0 / dh));
        double hp = 0.0, fp = 0.0;
        PO_TRY(k_reduce1(ctx, RED_DOT, xt->d, px->d, n, &hp));
        PO_TRY(k_reduce1(ctx, RED_DOT, gt->d, px->d, n, &fp));
        row("Hessian-vector product test (p^T H p)", -1, hp, fp);
      } else {
        *report += "Hessian-vector products are not implemented by this problem\n";
      }
    }
    return PO_OK;
  };
  if (rc == PO_OK) rc = run();
  vec_decref(g);
  vec_decref(gt);
  for (Vec *v : A) vec_decref(v);
  for (Vec *v : At) vec_decref(v);
  return rc;
}

bool Problem::sparseTransposeColumn(double alpha, Vec *x, Vec *pzw, GroupCol *col) {
  static const bool off = getenv("PAROPT_AMD_NO_GROUP_COLS") != nullptr;
  if (off || !grouped || nwcon <= 0 || gmap.start != 0 || nlocal >= 2000000000LL) return false;
  col->w = pzw->d;
  col->scale = alpha * group_alpha;  // the value k_group_scatter_set(..., alpha * group_alpha, ...) stores
  col->period = (unsigned)(gmap.nw + gmap.skip);
  col->nw = (unsigned)gmap.nw;
  col->nwcon = gmap.nwcon;
  return true;
}
bool Problem::sparseGramGroups(Vec *x, GramGroups *g) {
  if (!grouped || nwblock > 1 || nwcon <= 0) return false;
  g->nwcon = gmap.nwcon;
  g->start = gmap.start;
  g->nw = gmap.nw;
  g->skip = gmap.skip;
  g->alpha = group_alpha;  // as in sparseJacobianPanel
  return true;
}

int Problem::sparseApplyK0(Vec *x, Vec *d, Vec *cw, const double *bx, const double *bw, Vec *yx, Vec *yw,
                           Vec *wwork) {
  if (grouped && nwblock == 1) {
    // u = Aw (d o bx) in one tiled pass, yw = cw (bw - u), yx = d (bx + alpha yw[group]); one launch where the map tiles
    bool done = false;
    PO_TRY(k_group_k0(ctx, gmap, d->d, bx, cw->d, bw, group_alpha, nlocal, yx->d, yw->d, &done));
    if (done) return PO_OK;
    const double *P[1] = {bx};
    double *U[1] = {wwork->d};
    PO_TRY(k_group_panel(ctx, gmap, P, 1, d->d, group_alpha, U));
    PO_TRY(k_w_apply_mid(ctx, cw->d, bw, wwork->d, nwcon, yw->d));
    return k_group_apply(ctx, gmap, d->d, bx, group_alpha, yw->d, nlocal, yx->d);
  }
  if (csr) return csr->applyK0(d->d, bx, bw, yx->d, yw->d);
  const int64_t n = nlocal, w = nwcon;
  PO_TRY(k_mul(ctx, yx->d, 1.0, d->d, bx, n));
  if (bw) {
    PO_TRY(k_copy(ctx, yw->d, bw, w));
  } else {
    PO_TRY(k_fill(ctx, yw->d, w, 0.0));
  }
  if (addSparseJacobian(-1.0, x, yx, yw) != 0) return PO_ERR_USER;
  if (nwblock > 1) {  // applyFactor (:192-216)
    double *Y[1] = {yw->d};
    PO_TRY(k_blk_solve(ctx, blk->d, w / nwblock, nwblock, Y, 1, 0));
  } else {
    PO_TRY(k_mul(ctx, yw->d, 1.0, cw->d, yw->d, w));
  }
  PO_TRY(k_copy(ctx, yx->d, bx, n));
  if (addSparseJacobianTranspose(1.0, x, yw, yx) != 0) return PO_ERR_USER;
  PO_TRY(k_mul(ctx, yx->d, 1.0, d->d, yx->d, n));
  return PO_OK;
}

// ---- callbacks --------------------------------------------------------------------------------
int CallbackProblem::getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) {
  if (!cb.get_vars_and_bounds) {
    set_error("problem callbacks lack get_vars_and_bounds");
    return PO_ERR_ARG;
  }
  int rc = cb.get_vars_and_bounds(cb.user, static_cast<po_vec>(x), static_cast<po_vec>(lb),
                                  static_cast<po_vec>(ub));
  return rc;
}
int CallbackProblem::evalObjCon(Vec *x, double *fobj, double *cons) {
  if (csr) {  // ParOptSparseProblem::evalObjCon (.cpp:724-727)
    if (!csr_obj_con) return 1;
    return csr_obj_con(cb.user, static_cast<po_vec>(x), fobj, cons, static_cast<po_vec>(csr->cw));
  }
  return cb.eval_obj_con(cb.user, static_cast<po_vec>(x), fobj, cons);
}
int CallbackProblem::evalObjConGradient(Vec *x, Vec *g, Vec **Ac) {
  std::vector<po_vec> h(ncon > 0 ? ncon : 1);
  // Ac == nullptr with ncon > 0: objective gradient only (linear_constraints); with ncon == 0 the callback still
  // receives a valid (empty) array
  const bool only_g = (Ac == nullptr && ncon > 0);
  for (int j = 0; j < ncon; j++) h[j] = Ac ? static_cast<po_vec>(Ac[j]) : nullptr;
  if (csr) {  // ParOptSparseProblem::evalObjConGradient (.cpp:739-742)
    if (!csr_gradient) return 1;
    int rc = csr_gradient(cb.user, static_cast<po_vec>(x), static_cast<po_vec>(g), only_g ? nullptr : h.data(),
                          csr->data, csr->nnz);
    if (rc != 0) return rc;
    return csrValuesChanged() != PO_OK;
  }
  return cb.eval_obj_con_gradient(cb.user, static_cast<po_vec>(x), static_cast<po_vec>(g),
                                  only_g ? nullptr : h.data());
}
int CallbackProblem::computeQuasiNewtonUpdateCorrection(Vec *x, const double *z, Vec *s, Vec *y) {
  if (!cb.qn_update_correction) return 0;
  return cb.qn_update_correction(cb.user, static_cast<po_vec>(x), z, static_cast<po_vec>(s),
                                 static_cast<po_vec>(y));
}
int CallbackProblem::writeOutput(int iter, Vec *x) {
  if (!cb.write_output) return 0;
  return cb.write_output(cb.user, iter, static_cast<po_vec>(x));
}

int CallbackProblem::evalSparseCon(Vec *x, Vec *out) {
  if (csr) return Problem::evalSparseCon(x, out);
  if (!sparse.set || !sparse.cb.eval_sparse_con) return nwcon > 0 ? 1 : 0;
  return sparse.cb.eval_sparse_con(cb.user, static_cast<po_vec>(x), static_cast<po_vec>(out));
}
int CallbackProblem::addSparseJacobian(double alpha, Vec *x, Vec *px, Vec *out) {
  if (csr) return Problem::addSparseJacobian(alpha, x, px, out);
  if (!sparse.set || !sparse.cb.add_sparse_jacobian) return nwcon > 0 ? 1 : 0;
  return sparse.cb.add_sparse_jacobian(cb.user, alpha, static_cast<po_vec>(x),
                                       static_cast<po_vec>(px), static_cast<po_vec>(out));
}
int CallbackProblem::addSparseJacobianTranspose(double alpha, Vec *x, Vec *pzw, Vec *out) {
  if (csr) return Problem::addSparseJacobianTranspose(alpha, x, pzw, out);
  if (!sparse.set || !sparse.cb.add_sparse_jacobian_transpose) return nwcon > 0 ? 1 : 0;
  return sparse.cb.add_sparse_jacobian_transpose(cb.user, alpha, static_cast<po_vec>(x),
                                                 static_cast<po_vec>(pzw), static_cast<po_vec>(out));
}
int CallbackProblem::addSparseInnerProduct(double alpha, Vec *x, Vec *cvec, Vec *A) {
  if (csr) return Problem::addSparseInnerProduct(alpha, x, cvec, A);
  if (!sparse.set || !sparse.cb.add_sparse_inner_product) return nwcon > 0 ? 1 : 0;
  return sparse.cb.add_sparse_inner_product(cb.user, alpha, static_cast<po_vec>(x),
                                            static_cast<po_vec>(cvec), static_cast<po_vec>(A));
}

int CallbackProblem::evalHvecProduct(Vec *x, const double *z, Vec *zw, Vec *px, Vec *hvec) {
  if (!hvec_fn) return 1;
  return hvec_fn(cb.user, static_cast<po_vec>(x), z, static_cast<po_vec>(zw), static_cast<po_vec>(px),
                 static_cast<po_vec>(hvec));
}
int CallbackProblem::evalHessianDiag(Vec *x, const double *z, Vec *zw, Vec *hdiag) {
  if (!hdiag_fn) return 1;
  return hdiag_fn(cb.user, static_cast<po_vec>(x), z, static_cast<po_vec>(zw), static_cast<po_vec>(hdiag));
}

// ---- separable workloads ----------------------------------------------------------------------
SeparableProblem::SeparableProblem(Ctx *c, int kind_, int64_t nglobal_, int ncon_, uint64_t seed_,
                                   double eig_min_, double eig_max_)
    : Problem(c, 0, kind_ == PO_PROBLEM_ROSENBROCK ? 2 : ncon_,
              kind_ == PO_PROBLEM_ROSENBROCK ? 2 : ncon_),
      kind(kind_), seed(seed_), eig_min(eig_min_), eig_max(eig_max_), q(nullptr), b(nullptr) {
  nglobal = nglobal_;
  shard(nglobal_, c->rank, c->size, &nlocal, &offset);
}

SeparableProblem::~SeparableProblem() {
  vec_decref(chain_tmp);
  vec_decref(q);
  vec_decref(b);
  for (Vec *v : A) vec_decref(v);
}

int SeparableProblem::init() {
  const int64_t n = nlocal;
  beta.assign(ncon, 0.0);
  if (kind == PO_PROBLEM_ROSENBROCK) return PO_OK;
  b = vec_new(ctx, n);
  if (!b) return PO_ERR_HIP;
  PO_TRY(k_fill_hash(ctx, b->d, n, seed, 2, offset, 1.0, 0.0));
  if (kind == PO_PROBLEM_QUADRATIC) {
    q = vec_new(ctx, n);
    if (!q) return PO_ERR_HIP;
    PO_TRY(k_fill_hash(ctx, q->d, n, seed, 1, offset, eig_max - eig_min, eig_min));
  }
  for (int j = 0; j < ncon; j++) {
    Vec *a = vec_new(ctx, n);
    if (!a) return PO_ERR_HIP;
    PO_TRY(k_fill_hash(ctx, a->d, n, seed, 100 + j, offset, 1.0, 0.0));
    A.push_back(a);
  }
  if (kind == PO_PROBLEM_QUADRATIC) {
    // beta_j = u01(seed, 4, j): same hash on the host
    for (int j = 0; j < ncon; j++) {
      uint64_t z = seed * 0x9E3779B97F4A7C15ULL + 4 * 0xD1B54A32D192ED03ULL + (uint64_t)j;
      z += 0x9E3779B97F4A7C15ULL;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
      z = z ^ (z >> 31);
      beta[j] = (double)(z >> 11) * (1.0 / 9007199254740992.0);
    }
  } else {
    // beta_j = 0.25 * sum_i a_j[i]  (examples/random_convex/random_convex.py:110-111): the sums of
    // all rows in one pass, as dot products with a vector of ones
    Vec *ones = vec_new(ctx, n);
    if (!ones) return PO_ERR_HIP;
    PO_TRY(k_fill(ctx, ones->d, n, 1.0));
    std::vector<const double *> ap;
    for (Vec *v : A) ap.push_back(v->d);
    if (ncon > 0) PO_TRY(k_mdot(ctx, ones->d, ap.data(), ncon, n, beta.data()));
    for (int j = 0; j < ncon; j++) beta[j] *= 0.25;
    vec_decref(ones);
  }
  return PO_OK;
}

int SeparableProblem::setWeighting(int64_t nwg, int nw, int64_t nwstart, int nwskip, int64_t nwineq) {
  if (nwg < 0 || nw <= 0 || nwstart < 0 || nwskip < 0 || nwineq < 0 || nwineq > nwg) {
    set_error("setWeighting: bad arguments");
    return PO_ERR_ARG;
  }
  const int64_t period = (int64_t)nw + nwskip;
  if (nwstart + (nwg - 1) * period + nw > nglobal && nwg > 0) {
    set_error("setWeighting: groups reach past the last variable");
    return PO_ERR_ARG;
  }
  int64_t first = 0, count = nwg;  // global index of this rank's first group, number of local groups
  int64_t lstart = nwstart;
  if (ctx->size > 1) {
    // group i lives where its first variable lives; it must end on the same rank
    if (nwstart + nw > period || nlocal % period != 0 || offset % period != 0) {
      set_error("setWeighting: with %d ranks the shard size (%lld) must be a multiple of nw+nwskip",
                ctx->size, (long long)nlocal);
      return PO_ERR_ARG;
    }
    first = offset / period;
    count = nlocal / period;
    if (first > nwg) first = nwg;
    if (first + count > nwg) count = nwg - first;
  }
  gmap.nwcon = count;
  gmap.start = lstart;
  gmap.nw = nw;
  gmap.skip = nwskip;
  grouped = true;
  group_alpha = -1.0;  // Aw = -(group indicator)
  nwcon = count;
  nwinequality = nwineq - first;
  if (nwinequality < 0) nwinequality = 0;
  if (nwinequality > count) nwinequality = count;
  return PO_OK;
}
int SeparableProblem::setChain(int span, int stride, int reverse_cols) {
  if (span < 1 || stride < 1) {
    set_error("setChain: span and stride must be positive");
    return PO_ERR_ARG;
  }
  const int64_t rows = nlocal >= span ? (nlocal - span) / stride + 1 : 0;
  if (rows * span > 2000000000LL) {
    set_error("setChain: more than 2e9 Jacobian entries on one rank");
    return PO_ERR_ARG;
  }
  std::vector<int> rowp(rows + 1), cols((size_t)rows * span);
  for (int64_t i = 0; i < rows; i++) {
    rowp[i] = (int)(i * span);
    for (int k = 0; k < span; k++) {
      cols[i * span + (reverse_cols ? span - 1 - k : k)] = (int)(i * stride + k);
    }
  }
  rowp[rows] = (int)(rows * span);
  PO_TRY(setSparseJacobianData(rows, rows, rowp.data(), cols.data()));
  chain_span = span;
  chain_stride = stride;
  chain_reverse = reverse_cols;
  // (gmap is setSparseJacobianData's: a chain with stride >= span has row-disjoint groups, is recognised as the grouped
  // pattern and runs the group kernels for as long as its Jacobian entries -2 x are uniform -- a start at x = -1, say.
  // Until round 5's random campaign this function reset gmap behind the recognition: `grouped` with an empty map.)
  return PO_OK;
}
int SeparableProblem::evalSparseCon(Vec *x, Vec *out) {
  if (csr) return Problem::evalSparseCon(x, out);
  return k_group_sum(ctx, gmap, out->d, 0, 1.0, -1.0, x->d);  // cw_i = 1 - sum of the group
}
// Hessian of the Lagrangian f - z^T c - zw^T cw.  The weighting constraints are linear; a chain constraint
// cw_i = 1 - sum x^2 adds 2 zw_i on the diagonal entries of its variables.
int SeparableProblem::chainHessian(Vec *zw, Vec *px, Vec *h) {
  if (!csr || !zw) return PO_OK;
  if (!chain_tmp) chain_tmp = vec_new(ctx, nlocal);
  if (!chain_tmp) return PO_ERR_HIP;
  // a recognised chain (stride >= span, uniform entries) keeps the light pattern: no transposed index on the device.
  // Its rows are the groups of gmap, so the column sums are the group scatter (same value per entry: 2 zw_g).
  if (grouped || csr->light)
    PO_TRY(k_group_scatter_set(ctx, gmap, chain_tmp->d, 2.0, zw->d, nlocal));
  else
    PO_TRY(csr->colSum(2.0, zw->d, chain_tmp->d));
  if (px) PO_TRY(k_mul(ctx, chain_tmp->d, 1.0, chain_tmp->d, px->d, nlocal));
  return k_axpy(ctx, h->d, 1.0, chain_tmp->d, nlocal);
}
int SeparableProblem::evalHvecProduct(Vec *x, const double *z, Vec *zw, Vec *px, Vec *hvec) {
  if (kind == PO_PROBLEM_ROSENBROCK) {
    if (k_rosen_hess(ctx, x->d, z[0], px->d, nlocal, hvec->d) != PO_OK) return 1;
  } else if (k_sep_hess(ctx, kind == PO_PROBLEM_QUADRATIC ? 0 : 1, q ? q->d : nullptr, b->d, x->d, px->d, nlocal,
                        hvec->d) != PO_OK) {
    return 1;
  }
  return chainHessian(zw, px, hvec) != PO_OK;
}
int SeparableProblem::evalHessianDiag(Vec *x, const double *z, Vec *zw, Vec *hdiag) {
  if (kind == PO_PROBLEM_ROSENBROCK) {
    if (k_rosen_hess(ctx, x->d, z[0], nullptr, nlocal, hdiag->d) != PO_OK) return 1;
  } else if (k_sep_hess(ctx, kind == PO_PROBLEM_QUADRATIC ? 0 : 1, q ? q->d : nullptr, b->d, x->d, nullptr,
                        nlocal, hdiag->d) != PO_OK) {
    return 1;
  }
  return chainHessian(zw, nullptr, hdiag) != PO_OK;
}

int SeparableProblem::getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) {
  const int64_t n = nlocal;
  if (kind == PO_PROBLEM_QUADRATIC) {
    PO_TRY(k_fill_hash(ctx, x->d, n, seed, 3, offset, 1.0, -2.0));
    PO_TRY(k_fill(ctx, lb->d, n, -5.0));
    PO_TRY(k_fill(ctx, ub->d, n, 5.0));
  } else if (kind == PO_PROBLEM_CONVEX) {
    PO_TRY(k_fill_hash(ctx, x->d, n, seed, 3, offset, 0.9, 0.05));
    PO_TRY(k_fill(ctx, lb->d, n, 0.0));
    PO_TRY(k_fill(ctx, ub->d, n, 1.0));
  } else {
    PO_TRY(k_fill(ctx, x->d, n, -1.0));
    PO_TRY(k_fill(ctx, lb->d, n, -2.0));
    PO_TRY(k_fill(ctx, ub->d, n, 1.0));
  }
  return k_bounds_mode(ctx, x->d, lb->d, ub->d, bounds_mode, offset, n);
}

int SeparableProblem::evalObjCon(Vec *x, double *fobj, double *cons) {
  const int64_t n = nlocal;
  if (csr) PO_TRY(k_chain_con(ctx, x->d, nwcon, chain_span, chain_stride, csr->cw->d));
  if (kind == PO_PROBLEM_ROSENBROCK) {
    BatchScope batch(ctx);
    PO_TRY(k_rosen_f(ctx, x->d, n, rosen_out));
    return batch.end_then([this, fobj, cons] {
      *fobj = rosen_out[0];
      cons[0] = rosen_out[1] + 0.25;
      cons[1] = rosen_out[2] + 10.0;
    });
  }
  // objective sum and constraint products share one collective + host sync (and the caller's, if it has
  // reductions queued: the line search's barrier sums at the trial point)
  BatchScope batch(ctx);
  if (kind == PO_PROBLEM_QUADRATIC) {
    PO_TRY(k_quadratic_f(ctx, q->d, b->d, x->d, n, fobj));
  } else {
    PO_TRY(k_convex_f(ctx, b->d, x->d, n, fobj));
  }
  std::vector<const double *> ap;
  for (Vec *v : A) ap.push_back(v->d);
  if (ncon > 0) PO_TRY(k_mdot(ctx, x->d, ap.data(), ncon, n, cons));
  return batch.end_then([this, cons] {
    for (int j = 0; j < ncon; j++) {
      cons[j] = (kind == PO_PROBLEM_QUADRATIC) ? cons[j] + beta[j] : -cons[j] + beta[j];
    }
  });
}

int SeparableProblem::evalObjConGradient(Vec *x, Vec *g, Vec **Ac) {
  const int64_t n = nlocal;
  if (csr) {
    PO_TRY(k_chain_jac(ctx, x->d, nwcon, chain_span, chain_stride, chain_reverse, csr->data));
    PO_TRY(csrValuesChanged());
  }
  if (kind == PO_PROBLEM_ROSENBROCK) {
    if (!Ac) return 1;  // nonlinear constraints: the flag must not be set
    return k_rosen_g(ctx, x->d, n, g->d, Ac[0]->d, Ac[1]->d);
  }
  if (kind == PO_PROBLEM_QUADRATIC) {
    PO_TRY(k_quadratic_g(ctx, q->d, b->d, x->d, n, g->d));
  } else {
    PO_TRY(k_convex_g(ctx, b->d, x->d, n, g->d));
  }
  // the constraint Jacobian is rewritten at every gradient evaluation, as the reference's example
  // problems do (examples/random_convex/random_convex.py:67-71): Ac_j = +-a_j, one launch
  std::vector<double *> dst;
  std::vector<const double *> src;
  for (int j = 0; j < ncon && Ac; j++) {
    dst.push_back(Ac[j]->d);
    src.push_back(A[j]->d);
  }
  if (ncon > 0 && Ac) {
    PO_TRY(k_panel_lincomb(ctx, dst.data(), kind == PO_PROBLEM_QUADRATIC ? 1.0 : -1.0, src.data(), 0.0,
                           nullptr, ncon, n));
  }
  return PO_OK;
}

}  // namespace po
