// Interior-point iteration with every design-sized operation on the GPU (see ip.hpp).
//
// The control flow follows ParOptInteriorPoint::optimize (reference
// src/ParOptInteriorPoint.cpp:4399-5333) decision by decision; the linear algebra does NOT follow
// the reference's operation sequence but the fused form of DESIGN.md "Fused KKT step":
//   * ONE weighted Gram W = P^T diag(Dinv) P of the panel P = [Ac | Z] (MFMA) yields both Schur
//     complements G (:1932-1970) and Ce (:2634-2667) after O((c+k)^3) host work;
//   * every bordered solve K0^-1 b plus its Sherman-Morrison-Woodbury correction (:2700-2737) is
//     one panel-dot pass, tiny replicated host algebra, and one panel-axpy pass;
//   * iterative refinement (:4985-4991) folds the residual of the linearised system straight into
//     the right-hand side of the next solve;
//   * step scalings are applied lazily as scalars (no passes over the step vectors).
#include "ip.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>

namespace po {

// ================================================================================================
// construction
// ================================================================================================
static const int LS_SUCCESS = 1, LS_FAILURE = 2, LS_MIN_STEP = 4, LS_MAX_ITERS = 8,
                 LS_NO_IMPROVEMENT = 16, LS_SHORT_STEP = 32;  // src/ParOptInteriorPoint.h:220-225

InteriorPoint::InteriorPoint(Problem *p)
    : prob(p), ctx(p->ctx), n(p->nlocal), c(p->ncon), qn(nullptr), x(nullptr), zl(nullptr),
      zu(nullptr), lb(nullptr), ub(nullptr), g(nullptr), fobj(0.0), barrier_param(0.1),
      rho_penalty_search(0.0), niter(0), neval(0), ngeval(0), analytic_panel_dots(true),
      iter_cb(nullptr),
      iter_cb_user(nullptr), px(nullptr), pzl(nullptr), pzu(nullptr), Dinv(nullptr), rx(nullptr),
      tvec(nullptr), xt(nullptr), y_qn(nullptr), s_qn(nullptr), vA(nullptr), qn_created(false), qn_owned(true), wk(0),
      comp_prod(0), comp_count(0), max_rx(0), max_rzl(0), max_rzu(0), sx(1.0), sz(1.0),
      ptpx_valid(false), residual_fused(false), residual_cached(false), corrector_active(false),
      norm_type(0), tdots_valid(false), fused_dots(true), phase_t0(0) {
  qn_handle.qn = nullptr;
  hdiag = nullptr;
  vA_valid = false;
  inexact_newton_step = false;
  merit_cache_valid = false;
  fuse_merit = !getenv("PAROPT_AMD_NO_FUSED_MERIT");
  lean_step = !getenv("PAROPT_AMD_NO_LEAN_STEP");
  recompute_dt = !getenv("PAROPT_AMD_NO_RECOMPUTE_DT");
  nhvec = 0;
  nw = p->nwcon;
  has_w = false;
  nw_global = 0.0;
  gsw = gtw = Cw = wd2 = wyw = wtmp = wtmp2 = d1v = nullptr;
  for (int i = 0; i < 5; i++) wvar[i] = wresv[i] = wstepv[i] = wscalev[i] = nullptr;
  for (int i = 0; i < 10; i++) w_merit_last[i] = 0.0;
  for (int i = 0; i < 7; i++) w_sums[i] = 0.0;
  for (int i = 0; i < 5; i++) w_maxs[i] = 0.0;
  // debugging / test switch: re-measure P^T px with explicit mdot passes instead of W-based algebra
  if (getenv("PAROPT_AMD_EXPLICIT_DOTS")) analytic_panel_dots = false;
  if (getenv("PAROPT_AMD_NO_FUSED_DOTS")) fused_dots = false;
  fused_tdots = !getenv("PAROPT_AMD_NO_FUSED_TDOTS");
  recompute_first_step = !getenv("PAROPT_AMD_NO_RECOMPUTE");
  fuse_mult_update = !getenv("PAROPT_AMD_NO_FUSED_UPDATE");
  fast_yqn_w = !getenv("PAROPT_AMD_NO_FAST_YQN_W");
  w_lean = !getenv("PAROPT_AMD_NO_W_LEAN");
  spec_mu_on = !getenv("PAROPT_AMD_NO_SPEC_MU");
  recompute_rhs = recompute_first_step && !getenv("PAROPT_AMD_NO_RECOMPUTE_RHS");
  // Off by default: leaving the L-SR1 columns unformed saves the Gram pass 0.6 ms (its ten output streams) but costs
  // the two solve passes ten more input streams each, +1.2 ms at n = 50 M; kept as a switch.
  virtual_z = recompute_rhs && getenv("PAROPT_AMD_VIRTUAL_Z") != nullptr;
  use_acz = !getenv("PAROPT_AMD_NO_ACZ");
  use_ztpx_hint = !getenv("PAROPT_AMD_NO_ZTS_HINT");
  use_lower = prob->useLowerBounds();
  use_upper = prob->useUpperBounds();
  vars.resize(c);
  res.resize(c);
  step.resize(c);
  refine.resize(c);
  cvals.assign(c, 0.0);
  step_mins[0] = step_mins[1] = 1.0;
  setPenaltyGamma(options.real("penalty_gamma"));
}

int InteriorPoint::allocate() {
  Vec **all[] = {&x, &zl, &zu, &lb, &ub, &g, &px, &pzl, &pzu, &Dinv, &rx, &tvec, &xt, &y_qn, &s_qn, &vA};
  for (Vec **v : all) {
    *v = vec_new(ctx, n);
    if (!*v) return PO_ERR_HIP;
  }
  for (int j = 0; j < c; j++) {
    Vec *a = vec_new(ctx, n);
    if (!a) return PO_ERR_HIP;
    Ac.push_back(a);
  }
  // constructor state of the reference (:415-447): bounds checked once, multipliers = 1
  barrier_param = options.real("init_barrier_param");
  PO_TRY(initAndCheckDesignAndBounds());
  PO_TRY(k_fill(ctx, zl->d, n, 1.0));
  PO_TRY(k_fill(ctx, zu->d, n, 1.0));
  for (int i = 0; i < c; i++) vars.z[i] = vars.s[i] = vars.t[i] = vars.zs[i] = vars.zt[i] = 1.0;
  PO_TRY(allocateW());
  return PO_OK;
}

InteriorPoint::~InteriorPoint() {
  Vec *all[] = {x, zl, zu, lb, ub, g, px, pzl, pzu, Dinv, rx, tvec, xt, y_qn, s_qn, vA, acz};
  for (Vec *v : all) vec_decref(v);
  for (Vec *v : Ac) vec_decref(v);
  Vec *wall[] = {gsw, gtw, Cw, wd2, wyw, wtmp, wtmp2, d1v, cwx};
  for (Vec *v : wall) vec_decref(v);
  for (int i = 0; i < 5; i++) {
    vec_decref(wvar[i]);
    vec_decref(wresv[i]);
    vec_decref(wstepv[i]);
    vec_decref(wscalev[i]);
  }
  for (Vec *v : Uw) vec_decref(v);
  for (Vec *v : gmresW) vec_decref(v);
  vec_decref(hdiag);
  if (qn_owned) delete qn;
  if (user_events_ready)
    for (int i = 0; i < 2 * kUserRing; i++) (void)hipEventDestroy(user_ev[i]);
}

void InteriorPoint::setPenaltyGamma(double gamma) {  // :1127-1151
  if (gamma < 0.0) return;
  gamma_s.assign(c, gamma);
  gamma_t.assign(c, gamma);
  for (int i = 0; i < c && i < prob->ninequality; i++) gamma_s[i] = 0.0;
  if (has_w) k_w_gamma(ctx, gsw->d, gtw->d, gamma, prob->nwinequality, nw);  // :1139-1150
}

int InteriorPoint::setQuasiNewton(CompactQuasiNewton *qn_) {  // :1193-1234
  if (qn_owned) delete qn;
  qn = qn_;
  qn_owned = false;
  qn_created = true;
  qn_handle.qn = qn;
  return PO_OK;
}

int InteriorPoint::resetProblemInstance(Problem *p) {  // :745-764
  if (p->nlocal != prob->nlocal || p->ncon != prob->ncon || p->nwcon != prob->nwcon ||
      p->ninequality != prob->ninequality || p->nwinequality != prob->nwinequality) {
    set_error("ParOpt: Incompatible problem instance");
    return PO_ERR_ARG;
  }
  prob = p;
  ac_valid = false;
  acz_valid = false;
  cwx_valid = false;
  return PO_OK;
}

void InteriorPoint::setPenaltyGammaArray(const double *gamma) {  // :1160-1172 (dense blocks only)
  for (int i = 0; i < c; i++) {
    if (gamma[i] >= 0.0) {
      gamma_s[i] = i < prob->ninequality ? 0.0 : gamma[i];
      gamma_t[i] = gamma[i];
    }
  }
}

int InteriorPoint::createQuasiNewton() {  // ctor :262-292
  if (qn_created) return PO_OK;
  qn_created = true;
  const std::string qt = options.str("qn_type");
  const int msub = options.integer("qn_subspace_size");
  if (qt == "bfgs") {
    LBFGS *b = new LBFGS(ctx, n, msub);
    b->setBFGSUpdateType(std::string(options.str("qn_update_type")) == "damped_update"
                             ? PO_BFGS_DAMPED_UPDATE
                             : PO_BFGS_SKIP_NEGATIVE_CURVATURE);
    qn = b;
  } else if (qt == "sr1") {
    qn = new LSR1(ctx, n, msub);
  }
  if (qn) {
    const std::string dt = options.str("qn_diag_type");
    qn->setInitDiagonalType(dt == "yts_over_sts" ? PO_QN_YTS_OVER_STS : PO_QN_YTY_OVER_YTS);
  }
  qn_handle.qn = qn;
  return PO_OK;
}

Bounds InteriorPoint::bounds() const {
  Bounds b;
  b.x = x->d;
  b.lb = lb->d;
  b.ub = ub->d;
  b.zl = zl->d;
  b.zu = zu->d;
  b.max_bound = options.real("max_bound_value");
  b.use_lower = use_lower;
  b.use_upper = use_upper;
  return b;
}

std::vector<const double *> InteriorPoint::panel(bool use_qn, int *k) const {
  std::vector<const double *> p;
  for (Vec *a : Ac) p.push_back(a->d);
  *k = 0;
  if (qn && use_qn) {
    std::vector<const double *> z = qn->zPointers();
    *k = (int)z.size();
    p.insert(p.end(), z.begin(), z.end());
  }
  return p;
}

void InteriorPoint::phaseBegin() {
  phase_t0 = std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
void InteriorPoint::phaseEnd(const char *name) {
  // every phase ends in a host-synchronising reduction or is followed by one, so host wall time
  // attributes the stream work to the phase that issued it (documented in DESIGN.md)
  const double t1 =
      std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
  for (size_t i = 0; i < phase_names.size(); i++) {
    if (phase_names[i] == name) {
      phase_seconds[i] += t1 - phase_t0;
      phase_t0 = t1;
      return;
    }
  }
  phase_names.push_back(name);
  phase_seconds.push_back(t1 - phase_t0);
  phase_t0 = t1;
}

void InteriorPoint::userBegin() {
  ctx->in_user = 1;
  if (!user_timing) return;
  if (!user_events_ready) {
    for (int i = 0; i < 2 * kUserRing; i++)
      if (hipEventCreate(&user_ev[i]) != hipSuccess) return;
    user_events_ready = true;
  }
  if (user_pending == kUserRing) userHarvest();
  (void)hipEventRecord(user_ev[2 * user_pending], ctx->stream);
}
void InteriorPoint::userEnd() {
  ctx->in_user = 0;
  if (!user_timing || !user_events_ready) return;
  (void)hipEventRecord(user_ev[2 * user_pending + 1], ctx->stream);
  user_pending++;
}
void InteriorPoint::userHarvest() {
  for (int i = 0; i < user_pending; i++) {
    float ms = 0.0f;
    if (hipEventSynchronize(user_ev[2 * i + 1]) == hipSuccess &&
        hipEventElapsedTime(&ms, user_ev[2 * i], user_ev[2 * i + 1]) == hipSuccess)
      user_seconds += 1e-3 * ms;
  }
  user_pending = 0;
}

// ================================================================================================
// bounds / starting point
// ================================================================================================
int InteriorPoint::initAndCheckDesignAndBounds() {  // :4277-4361
  int rc = prob->getVarsAndBounds(x, lb, ub);
  if (rc != 0) {
    set_error("getVarsAndBounds failed with code %d", rc);
    return PO_ERR_USER;
  }
  const double rel_bound = 0.001 * barrier_param;
  int flag = 0;
  PO_TRY(k_check_bounds(ctx, x->d, lb->d, ub->d, zl->d, zu->d, options.real("max_bound_value"),
                        rel_bound, use_lower && use_upper, n, &flag));
  check_flag |= flag;
  if (ctx->rank == 0) {
    if (flag & 1) history += "ParOpt Warning: Variable bounds are inconsistent\n";
    if (flag & 2) history += "ParOpt Warning: Variables may be too close to lower bound\n";
    if (flag & 4) history += "ParOpt Warning: Variables may be too close to upper bound\n";
  }
  return PO_OK;
}

int InteriorPoint::clampCounts(double out[8]) {
  const double eps = options.real("design_precision");
  PO_TRY(k_clamp_count(ctx, x->d, lb->d, ub->d, use_lower ? zl->d : nullptr, use_upper ? zu->d : nullptr, eps, n,
                       out));
  for (int j = 4; j < 8; j++) out[j] = 0.0;
  for (int i = 0; i < c; i++) {
    if (vars.s[i] == eps) out[4] += 1.0;
    if (vars.t[i] == eps) out[5] += 1.0;
    if (vars.zs[i] == eps) out[6] += 1.0;
    if (vars.zt[i] == eps) out[7] += 1.0;
  }
  return PO_OK;
}

// A host mirror the caller still holds for one of the solver's vectors (getOptimizedPoint + getArray) is the vector's
// data while it is live: after an internal writer has changed the device copy the mirror is refreshed, so that the
// caller's pointer shows the new values (the reference has one buffer) and optimize()'s upload of live mirrors cannot
// put the old ones back.
static int refreshLiveMirror(Vec *v) {
  if (!v || !v->h_live || !v->h || v->n <= 0) return PO_OK;
  PO_HIP(hipMemcpyAsync(v->h, v->d, sizeof(double) * (size_t)v->n, hipMemcpyDeviceToHost, v->ctx->stream));
  PO_HIP(hipStreamSynchronize(v->ctx->stream));
  return PO_OK;
}

int InteriorPoint::resetDesignAndBounds() {  // :1249-1251
  cwx_valid = false;
  int rc = prob->getVarsAndBounds(x, lb, ub);
  if (rc != 0) return PO_ERR_USER;
  PO_TRY(refreshLiveMirror(x));
  PO_TRY(refreshLiveMirror(lb));
  return refreshLiveMirror(ub);
}

void InteriorPoint::resetQuasiNewtonHessian() {
  if (qn) qn->reset();
}

int InteriorPoint::initLeastSquaresMultipliers() {  // :5366-5534 (w = 0)
  acz_valid = false;
  const double mu0 = options.real("init_barrier_param");
  PO_TRY(k_fill(ctx, zl->d, n, mu0));
  PO_TRY(k_fill(ctx, zu->d, n, mu0));
  for (int i = 0; i < c; i++) vars.z[i] = vars.s[i] = vars.t[i] = vars.zs[i] = vars.zt[i] = mu0;
  PO_TRY(k_zero_inactive(ctx, lb->d, ub->d, zl->d, zu->d, options.real("max_bound_value"), n));
  if (has_w) return initLeastSquaresMultipliersW();
  if (c == 0) return PO_OK;
  const double small = 1e-4;
  PO_TRY(k_fill(ctx, Dinv->d, n, 1.0));
  int k = 0;
  std::vector<const double *> A = panel(false, &k);
  std::vector<double> G((size_t)c * c, 0.0);
  PO_TRY(k_wgram(ctx, Dinv->d, A.data(), c, n, G.data()));
  for (int i = 0; i < c; i++) G[(size_t)i * (c + 1)] += small;
  std::vector<int> piv(c);
  lu_factor(c, G.data(), c, piv.data());
  // rhs = -(g - zl + zu); z = G^-1 ( -A rhs )
  const double al[2] = {1.0, -1.0};
  const double *vv[2] = {zl->d, zu->d};
  PO_TRY(k_panel_axpy(ctx, tvec->d, -1.0, g->d, 0.0, al, vv, 2, n));
  std::vector<double> z(c, 0.0);
  PO_TRY(k_mdot(ctx, tvec->d, A.data(), c, n, z.data()));
  for (int i = 0; i < c; i++) z[i] = -z[i];
  lu_solve(c, G.data(), c, piv.data(), z.data());
  for (int i = 0; i < c; i++) {
    const double gam = 10.0 * std::max(gamma_s[i], gamma_t[i]);
    vars.z[i] = (z[i] < -gam || z[i] > gam) ? 0.0 : z[i];
  }
  return PO_OK;
}

int InteriorPoint::initAffineStepMultipliers() {  // :5536-5656
  const double amin = options.real("start_affine_multiplier_min");
  PO_TRY(initLeastSquaresMultipliers());
  PO_TRY(k_zero_inactive(ctx, lb->d, ub->d, zl->d, zu->d, options.real("max_bound_value"), n));
  PO_TRY(computeResidual(0.0, true));
  bool use_qn = !(options.integer("sequential_linear_method") ||
                  !options.integer("use_qn_gmres_precon") || options.integer("use_diag_hessian"));
  if (options.integer("use_diag_hessian")) PO_TRY(ensureHdiag());
  PO_TRY(setUpKKTSystem(use_qn));
  denseResidual(0.0, res);
  if (has_w) {
    PO_TRY(solveKKTW(res, 0.0, use_qn, false, 1.0, step));
    PO_TRY(k_w_affine(ctx, wv(), wp(), amin, nw));  // :5601-5628
  } else {
    PO_TRY(solveKKT(res, 0.0, use_qn, false, 1.0, step));
  }
  acz_valid = false;
  for (int i = 0; i < c; i++) {
    vars.z[i] = vars.z[i] + step.z[i];
    vars.s[i] = std::max(amin, fabs(vars.s[i] + step.s[i]));
    vars.t[i] = std::max(amin, fabs(vars.t[i] + step.t[i]));
    vars.zs[i] = std::max(amin, fabs(vars.zs[i] + step.zs[i]));
    vars.zt[i] = std::max(amin, fabs(vars.zt[i] + step.zt[i]));
  }
  PO_TRY(k_affine_mult(ctx, bounds(), zl->d, pzl->d, zu->d, pzu->d, amin, n));
  double comp = 0.0;
  PO_TRY(getComplementarity(&comp));
  barrier_param = comp;
  return PO_OK;
}

// ================================================================================================
// residuals and norms
// ================================================================================================
void InteriorPoint::denseResidual(double mu, Dense &r) const {  // :1403-1409
  for (int i = 0; i < c; i++) {
    r.z[i] = -(cvals[i] - vars.s[i] + vars.t[i]);
    r.s[i] = -(gamma_s[i] - vars.zs[i] + vars.z[i]);
    r.t[i] = -(gamma_t[i] - vars.zt[i] - vars.z[i]);
    r.zs[i] = -(vars.s[i] * vars.zs[i] - mu);
    r.zt[i] = -(vars.t[i] * vars.zt[i] - mu);
  }
}

// the barrier parameter the monotone strategy switches to from the current one (:4693-4707)
double InteriorPoint::nextMonotoneMu() const {
  const double abs_res_tol = options.real("abs_res_tol");
  const double mu_frac = options.real("monotone_barrier_fraction") * barrier_param;
  const double mu_pow = pow(barrier_param, options.real("monotone_barrier_power"));
  double new_mu = mu_frac;
  if (mu_pow < mu_frac) new_mu = mu_pow;
  if (new_mu < 0.1 * abs_res_tol) new_mu = 0.09999 * abs_res_tol;
  return new_mu;
}

int InteriorPoint::computeResidual(double mu, bool vectors, Vec *yqn_complete, const MultUpdate *upd) {
  const double beta_mu = options.real("rel_bound_barrier") * mu;
  double *out = res_out;  // a member: inside a BatchScope the values arrive at the flush (after_reduce below)
  // speculative maxima for the next barrier parameter (see ip.hpp): only beside the norms of the CURRENT one
  const bool spec = vectors && spec_enabled && spec_mu_on && !has_w && mu == barrier_param;
  const double mu2 = spec ? nextMonotoneMu() : 0.0;
  const double beta_mu2 = spec ? options.real("rel_bound_barrier") * mu2 : -1.0;
  spec_valid = false;
  spec_dt_valid = false;  // (rx is about to change: a Dinv / t pair left by an earlier pass is stale)
  // sparse and design blocks of the residual share one collective + sync (the problem's sparse callbacks run in
  // between: built-in problems only)
  BatchScope wbatch(ctx, has_w && prob->reductionsBatchable());
  if (has_w) PO_TRY(computeResidualW(mu));
  bool acz_mode = false, acz_rebuilt = false;
  if (vectors) {
    std::vector<const double *> A;
    std::vector<double> zc;
    // A problem that declared its dense constraints linear (constant Jacobian): A^T z is kept in `acz` and
    // follows the multiplier steps (computeStepAndUpdate), so the residual does not stream the c constraint
    // gradients; it is rebuilt from the gradients every kAczRefresh uses to bound the round-off drift.
    if (prob->linear_constraints && ac_valid && c > 0 && use_acz) {
      if (!acz) acz = vec_new(ctx, n);
      if (!acz) return PO_ERR_HIP;
      if (!acz_valid || acz_age >= kAczRefresh) {
        std::vector<const double *> Ap;
        for (Vec *a : Ac) Ap.push_back(a->d);
        PO_TRY(k_panel_axpy(ctx, acz->d, 0.0, nullptr, 0.0, vars.z.data(), Ap.data(), c, n));
        acz_valid = true;
        acz_age = 0;
        acz_rebuilt = true;
      }
      acz_age++;
      A.push_back(acz->d);
      zc.push_back(1.0);
      acz_mode = true;
    } else {
      for (Vec *a : Ac) A.push_back(a->d);
      zc = vars.z;
    }
    GroupCol gcol;  // + Aw^T zw as one more panel column (:1358-1361): described (structured problems) or stored
    const bool have_gcol = has_w && !acz_mode && prob->sparseTransposeColumn(1.0, x, wvar[0], &gcol);
    if (has_w && !have_gcol) {
      if (prob->setSparseJacobianTranspose(1.0, x, wvar[0], tvec) != 0) return PO_ERR_USER;
      A.push_back(tvec->d);
      zc.push_back(1.0);
    }
    if (upd) {
      // the bound multipliers take their step in the same pass (see kkt_res_update_kernel); A^T z follows the dense
      // multiplier step by recurrence unless it has just been rebuilt from the new multipliers
      const double az_acz = (acz_mode && upd->acz_follow && !acz_rebuilt) ? upd->az : 0.0;
      // (lean step: px and the old point -- xt after the swap of computeStepAndUpdate -- instead of pzl / pzu)
      // No quasi-Newton update follows (a fixed approximation, the sequential linear method): the diagonal of the next
      // KKT system is known already, so this pass also leaves Dinv and t = Dinv o d1 of the next first solve behind
      // (spec_dt_*: setUpKKTSystem skips its pass over the bound data when diagonal and barrier parameter still match)
      const bool spec_dt = spec_dt_want && !has_w && fused_tdots && !options.integer("use_diag_hessian") &&
                           dbg_switch(SW_SPEC_DT, "PAROPT_AMD_SPEC_DT", 1) != 0;
      const double sdiag = spec_dt ? options.real("qn_sigma") + ((qn && !options.integer("sequential_linear_method"))
                                                                    ? qn->diag() : 0.0) : 0.0;
      const double sbmu = spec_dt ? options.real("rel_bound_barrier") * spec_dt_mu : 0.0;
      PO_TRY(k_kkt_res_update(ctx, bounds(), g->d, A.data(), zc.data(), acz_mode ? 0 : (int)A.size(), beta_mu, n,
                              rx->d, out, yqn_complete ? yqn_complete->d : nullptr, zl->d, pzl->d, zu->d, pzu->d,
                              upd->a, upd->eps, (yqn_complete || az_acz != 0.0) ? vA->d : nullptr,
                              upd->az, acz_mode ? acz->d : nullptr, az_acz, pz_stored ? nullptr : px->d,
                              pz_stored ? nullptr : xt->d, step_beta_mu, beta_mu2, have_gcol ? &gcol : nullptr, 1.0,
                              spec_dt ? Dinv->d : nullptr, spec_dt ? tvec->d : nullptr, sdiag, sbmu));
      if (spec_dt) {
        spec_dt_valid = true;
        spec_dt_diag = sdiag;
        spec_dt_bmu = sbmu;
      }
    } else {
      PO_TRY(k_kkt_res(ctx, bounds(), g->d, A.data(), zc.data(), (int)A.size(), beta_mu, n, rx->d, out,
                       yqn_complete ? yqn_complete->d : nullptr, beta_mu2, have_gcol ? &gcol : nullptr, 1.0));
    }
  } else {
    PO_TRY(k_res_norms(ctx, bounds(), beta_mu, n, out));
  }
  after_reduce(ctx, [this, vectors, spec, mu2] {
    const double *o = res_out;
    if (vectors) {
      l1_rx = o[2];
      l2_rx = o[5];
      max_rx = o[8];
    }
    if (spec) {
      spec_mu = mu2;
      spec_max[0] = o[11];
      spec_max[1] = o[12];
      spec_valid = true;
    }
    comp_prod = o[0];
    comp_count = o[1];
    l1_rzl = o[3];
    l1_rzu = o[4];
    l2_rzl = o[6];
    l2_rzu = o[7];
    max_rzl = o[9];
    max_rzu = o[10];
  });
  // (inside computeStepAndUpdate's batch the norms arrive with the quasi-Newton products: everything above that reads
  // them runs through after_reduce)
  return wbatch.end_nested();
}

void InteriorPoint::resNorms(const Dense &r, double *max_prime, double *max_dual,
                             double *max_infeas, double *res_norm) const {  // :1588-1723
  double mp = 0.0, md = 0.0, mi = 0.0;
  if (norm_type == 0) {  // infinity
    mp = max_rx;
    for (int i = 0; i < c; i++) {
      mp = std::max(mp, std::max(fabs(r.s[i]), fabs(r.t[i])));
      mi = std::max(mi, fabs(r.z[i]));
      md = std::max(md, std::max(fabs(r.zs[i]), fabs(r.zt[i])));
    }
    if (use_lower) md = std::max(md, max_rzl);
    if (use_upper) md = std::max(md, max_rzu);
    if (has_w) {  // :1599-1612
      mi = std::max(mi, w_maxs[0]);
      md = std::max(md, std::max(std::max(w_maxs[1], w_maxs[2]), std::max(w_maxs[3], w_maxs[4])));
    }
  } else if (norm_type == 1) {  // l1
    mp = l1_rx;
    for (int i = 0; i < c; i++) {
      mp += fabs(r.s[i]);
      mp += fabs(r.t[i]);
      mi += fabs(r.z[i]);
      md += fabs(r.zs[i]);
      md += fabs(r.zt[i]);
    }
    if (use_lower) md += l1_rzl;
    if (use_upper) md += l1_rzu;
    if (has_w) {
      mi += w_sums[1];
      md += w_sums[3] + w_sums[4] + w_sums[5] + w_sums[6];
    }
  } else {  // l2
    double prime = 0.0, infeas = 0.0, dual = 0.0;
    for (int i = 0; i < c; i++) {
      prime += r.s[i] * r.s[i] + r.t[i] * r.t[i];
      infeas += r.z[i] * r.z[i];
      dual += r.zs[i] * r.zs[i] + r.zt[i] * r.zt[i];
    }
    mp = l2_rx + prime;
    mi = infeas;
    md = dual;
    if (use_lower) md += l2_rzl;
    if (use_upper) md += l2_rzu;
    if (has_w) {  // the reference squares the l1 norms of the sparse dual blocks here (:1633-1638)
      mi += w_sums[2];
      md += w_sums[3] * w_sums[3] + w_sums[4] * w_sums[4] + w_sums[5] * w_sums[5] +
            w_sums[6] * w_sums[6];
    }
    mp = sqrt(mp);
    mi = sqrt(mi);
    md = sqrt(md);
  }
  *max_prime = mp;
  *max_dual = md;
  *max_infeas = mi;
  *res_norm = std::max(mp, std::max(md, mi));
}

double InteriorPoint::compFromSums(double prod, double count, const Dense &v,
                                   double wprod) const {  // :2742-2820
  prod = prod / options.real("rel_bound_barrier");
  if (has_w) {
    prod += wprod;
    count += 2.0 * nw_global;
  }
  for (int i = 0; i < c; i++) {
    prod += v.s[i] * v.zs[i] + v.t[i] * v.zt[i];
    count += 2.0;
  }
  return count != 0.0 ? prod / count : 0.0;
}

int InteriorPoint::getComplementarity(double *comp) {
  double out[11];
  PO_TRY(k_res_norms(ctx, bounds(), 0.0, n, out));
  double wprod = 0.0;
  if (has_w) PO_TRY(wCompStep(0.0, 0.0, &wprod));
  *comp = compFromSums(out[0], out[1], vars, wprod);
  return PO_OK;
}

// ================================================================================================
// the KKT system
// ================================================================================================
int InteriorPoint::setUpKKTSystem(bool use_qn, bool diag_only, const double *rhs_mu) {  // setUpKKTDiagSystem + setUpKKTSystem
  ptpx_valid = false;
  t0_valid = false;
  const double sigma = options.real("qn_sigma");
  const bool use_hdiag = options.integer("use_diag_hessian") && hdiag;  // h_i replaces b0 (:1840-1842)
  const double b0 = (!use_hdiag && qn && (use_qn || diag_only)) ? qn->diag() : 0.0;
  // The right-hand side of the solve that follows is known already (rx, the bounds and mu): t = Dinv o d1 is
  // built now (in the same pass over the bound data as Dinv) and rides through the Gram pass as one more,
  // pre-weighted, column, so that P^T t -- the mdot pass at the head of solveKKT -- comes out of the pass over P
  // that the Schur complements need anyway.
  const bool fuse_t = rhs_mu && fused_tdots && !has_w && c + (qn && use_qn && !diag_only ? qn->size() : 0) > 0;
  t_is_plain_dinv_d1 = false;
  // sparse constraints: the RAW right-hand side d1 of the first solve (the block solve applies to it) comes out of
  // the same pass over the bound data as Dinv (round 4; it is used below when the fused first solve is taken)
  const bool raw_d1_w = has_w && rhs_mu && fused_tdots && !corrector_active;
  const bool have_spec_dt = spec_dt_valid;
  spec_dt_valid = false;  // consumed here, or overwritten below
  if (fuse_t && have_spec_dt && !use_hdiag && spec_dt_diag == b0 + sigma &&
      spec_dt_bmu == options.real("rel_bound_barrier") * (*rhs_mu)) {
    // the residual pass of this iterate left exactly this Dinv and t behind (kkt_res_update_kernel)
    t_is_plain_dinv_d1 = true;
    t0_diag = b0 + sigma;
  } else if (fuse_t) {
    PO_TRY(k_dinv_d1(ctx, bounds(), b0 + sigma, use_hdiag ? hdiag->d : nullptr, rx->d,
                     options.real("rel_bound_barrier") * (*rhs_mu), n, Dinv->d, tvec->d));
    t_is_plain_dinv_d1 = !use_hdiag;
    t0_diag = b0 + sigma;
  } else if (raw_d1_w) {
    PO_TRY(k_dinv_d1(ctx, bounds(), b0 + sigma, use_hdiag ? hdiag->d : nullptr, rx->d,
                     options.real("rel_bound_barrier") * (*rhs_mu), n, Dinv->d, d1v->d, 1));
  } else {
    PO_TRY(k_dinv(ctx, bounds(), b0 + sigma, n, Dinv->d, use_hdiag ? hdiag->d : nullptr));
  }
  int k = 0;
  // L-SR1 leaves its columns Z_j = Y_j - b0 S_j unformed after an update; when this Gram pass is their
  // first consumer they are formed on the fly (and written out for the later passes): the panel is
  // ordered [Z | Ac] for that launch and the result permuted back to [Ac | Z]
  std::vector<const double *> Yp, Sp;
  std::vector<double *> Zo;
  double b0z = 0.0;
  const bool fuse_z = qn && use_qn && !diag_only && !has_w && qn->pendingZ(&Yp, &Sp, &Zo, &b0z) &&
                      !Yp.empty() && Yp.size() <= 12 && c + (int)Yp.size() + (fuse_t ? 1 : 0) <= kWgramMaxVecs;
  if (fuse_z) {
    k = (int)Yp.size();
    const int m2 = c + k, mt = m2 + (fuse_t ? 1 : 0);
    std::vector<const double *> P2(Yp);
    for (Vec *a : Ac) P2.push_back(a->d);
    if (fuse_t) P2.push_back(tvec->d);
    std::vector<double> W2((size_t)mt * mt, 0.0);
    // Leave the columns unformed when the solves that follow take them unformed too (solveKKT): the ten output
    // streams of the formation cost as much as forty input streams.  Any other consumer forms them on demand
    // (CompactQuasiNewton::zPointers).
    const bool leave_unformed = virtual_z && allow_virtual_z && fuse_t && k <= kMaxVirt;
    std::vector<double *> Znull(Zo.size(), nullptr);
    PO_TRY(k_wgram(ctx, Dinv->d, P2.data(), mt, n, W2.data(), Sp.data(), leave_unformed ? Znull.data() : Zo.data(), k,
                   b0z, fuse_t ? 1 : 0));
    if (!leave_unformed) qn->pendingZDone();
    W.assign((size_t)m2 * m2, 0.0);
    auto perm = [&](int i) { return i < c ? k + i : i - c; };  // index in [Ac | Z] -> index in [Z | Ac]
    for (int j = 0; j < m2; j++)
      for (int i = 0; i < m2; i++) W[i + (size_t)m2 * j] = W2[perm(i) + (size_t)mt * perm(j)];
    if (fuse_t) {
      t0dots.assign(m2, 0.0);
      for (int i = 0; i < m2; i++) t0dots[i] = W2[perm(i) + (size_t)mt * m2];
    }
  }
  std::vector<const double *> P;
  if (!fuse_z) {
    P = panel(use_qn && !diag_only, &k);  // (forms unformed columns: the plain Gram launch below reads them)
  } else if (has_w) {
    int k2 = 0;
    P = panel(use_qn && !diag_only, &k2);
  }
  const int m = c + k;
  wk = k;
  // Sparse constraints: the right-hand side of the first solve goes through the quasi-definite block solve,
  // t = [K0^-1 (d1, d2)]_x, which needs Cw -- so the factor comes first, then t (and the sparse multiplier part
  // wyw), then the Gram pass with t as its pre-weighted last column, as on the dense path.
  bool fuse_tw = false;
  // sparse constraints: the sparse residual norms of the new barrier parameter, the Gram pass and its sparse correction
  // share ONE collective + host sync (three before); the host algebra on W waits for the flush below
  BatchScope wbatch(ctx, has_w && prob->reductionsBatchable());
  bool t0_ready = fuse_z && fuse_t;
  if (has_w) {  // Cw = 1/(sw/zsw + tw/ztw + Aw Dinv Aw^T) (:1912-1930)
    PO_TRY(prob->sparseFactorFromSlacks(x, Dinv, wv(), Cw));  // Cdiag (:1912-1927) + mat->factor (:1930)
    fuse_tw = rhs_mu && fused_tdots && m > 0 && m + 1 <= kWgramMaxVecs && !corrector_active;
    if (fuse_tw) {
      PO_TRY(computeResidualW(*rhs_mu, true, true));  // ... with d2 of the block solve below (wd2)
      if (!raw_d1_w)
        PO_TRY(k_d1(ctx, bounds(), rx->d, nullptr, options.real("rel_bound_barrier") * (*rhs_mu), n, d1v->d));
      PO_TRY(applyK0(d1v->d, wd2->d, tvec, wyw));
    }
  }
  // structured sparse constraints: the panel image U = Aw (Dinv o P) of the Gram correction rides in the Gram pass
  // itself (one pass over the panel instead of two) where the problem and the kernel allow it
  GramGroups gram_groups;
  std::vector<double *> Ugs;
  const GramGroups *ggp = nullptr;
  bool panel_done = false;
  if (has_w && !fuse_z && m > 0 && prob->sparseGramGroups(x, &gram_groups)) {
    PO_TRY(panelImageVectors(m, Ugs));
    gram_groups.ncols = m;
    gram_groups.U = Ugs.data();
    ggp = &gram_groups;
  }
  if (!fuse_z) {
    W.assign((size_t)m * m, 0.0);
    if (m > 0 && (fuse_t || fuse_tw) && m + 1 <= kWgramMaxVecs) {
      std::vector<const double *> Pt(P);
      Pt.push_back(tvec->d);
      const int mt = m + 1;
      Wt_buf.assign((size_t)mt * mt, 0.0);
      PO_TRY(k_wgram(ctx, Dinv->d, Pt.data(), mt, n, Wt_buf.data(), nullptr, nullptr, 0, 0.0, 1, wbatch.open, ggp,
                     &panel_done));
      t0dots.assign(m, 0.0);
      after_reduce(ctx, [this, m, mt] {
        for (int j = 0; j < m; j++)
          for (int i = 0; i < m; i++) W[i + (size_t)m * j] = Wt_buf[i + (size_t)mt * j];
        for (int i = 0; i < m; i++) t0dots[i] = Wt_buf[i + (size_t)mt * m];
      });
      t0_ready = true;
    } else {
      if (m > 0)
        PO_TRY(k_wgram(ctx, Dinv->d, P.data(), m, n, W.data(), nullptr, nullptr, 0, 0.0, 0, wbatch.open, ggp,
                       &panel_done));
      t0dots.clear();
    }
  }
  if ((fuse_t || fuse_tw) && m > 0 && t0_ready && (int)t0dots.size() == m) {
    t0_valid = true;
    t0_mu = *rhs_mu;
  }
  // W -= U^T Cw U (d1v is free again: scratch of the panel image; tvec holds t)
  if (has_w) PO_TRY(sparseGramCorrection(P, m, fuse_tw ? d1v : tvec, wbatch.open, panel_done));
  PO_TRY(wbatch.end());
  // test aid: a relative perturbation of 1e-9 in one Gram entry, which the known-answer tests must detect
  // (tests/test_gpu_kat.py::test_kat_detects_a_perturbed_gram)
  // -- reachable only through po_debug_set_switch and only inside po_ip_debug_kkt (no environment variable)
  if (m > 1 && debug_keep_schur && dbg_switch(SW_PERTURB_W, nullptr, 0) != 0) {
    W[1] *= 1.0 + 1e-9;
    W[m] = W[1];
  }
  // G = W_AA + diag(s/zs + t/zt)   (:1952-1970)
  Gf.assign((size_t)c * c, 0.0);
  gpiv.assign(c, 0);
  for (int j = 0; j < c; j++)
    for (int i = 0; i < c; i++) Gf[i + (size_t)c * j] = W[i + (size_t)m * j];
  for (int i = 0; i < c; i++) Gf[(size_t)i * (c + 1)] += vars.s[i] / vars.zs[i] + vars.t[i] / vars.zt[i];
  if (debug_keep_schur) Gmat0 = Gf;  // as assembled, for po_ip_debug_kkt only
  if (c > 0) lu_factor(c, Gf.data(), c, gpiv.data());
  // Ce = W_ZZ - W_ZA G^-1 W_AZ - M / (d0 d0^T)   (:2634-2667 via SURVEY.md 3.4)
  Cef.clear();
  cpiv.clear();
  if (k > 0) {
    const double *d0, *M;
    double b0_;
    qn->getCompactMat(&b0_, &d0, &M, nullptr);
    Cef.assign((size_t)k * k, 0.0);
    cpiv.assign(k, 0);
    std::vector<double> col(c > 0 ? c : 1);
    for (int j = 0; j < k; j++) {
      for (int i = 0; i < c; i++) col[i] = W[i + (size_t)m * (c + j)];  // W_AZ[:, j]
      if (c > 0) lu_solve(c, Gf.data(), c, gpiv.data(), col.data());
      for (int i = 0; i < k; i++) {
        double v = W[(c + i) + (size_t)m * (c + j)];
        for (int l = 0; l < c; l++) v -= W[(c + i) + (size_t)m * l] * col[l];
        v -= M[i + (size_t)k * j] / (d0[i] * d0[j]);
        Cef[i + (size_t)k * j] = v;
      }
    }
    if (debug_keep_schur) Ce0 = Cef;
    lu_factor(k, Cef.data(), k, cpiv.data());
  } else {
    Ce0.clear();
  }
  return PO_OK;
}

// One application of [K0 + quasi-Newton correction]^-1 (computeKKTStep :2700-2737 with both
// solveKKTDiagSystem overloads :2074-2369 folded together).
//   first pass : b is the residual (rx on the device, dense blocks in `b`)  -> writes px, pzl, pzu
//   refine pass: tvec already holds Dinv*d1' (k_res_step)                    -> accumulates
int InteriorPoint::solveKKT(const Dense &b, double mu, bool use_qn, bool refine_pass, double tau,
                            Dense &out, bool fuse_residual) {
  const double beta_mu = options.real("rel_bound_barrier") * mu;
  const int k = (qn && use_qn) ? qn->size() : 0;
  if (k != wk) {
    set_error("internal: panel width changed between setUpKKTSystem and solve (%d vs %d)", k, wk);
    return PO_ERR_ARG;
  }
  const int m = c + k;
  // The panel [Ac | Z] is only asked for when a pass needs it: zPointers() FORMS unformed L-SR1 columns (one pass
  // over 3k vectors, k of them written), and the fused passes below take them unformed (Y_j - b0 S_j in registers).
  std::vector<const double *> P;
  bool have_panel = false;
  auto need_panel = [&]() {
    if (!have_panel) {
      int k2 = 0;
      P = panel(use_qn, &k2);
      have_panel = true;
    }
  };
  const bool corr = corrector_active && !refine_pass;
  // (corrector_fused: the corrector products are formed inside the two passes below from the affine step in
  // (px, pzl, pzu), nothing was stored in s_qn / y_qn)
  const bool corr_fused = corr && corrector_fused && m >= 1 && m <= kCorrDotsMax;
  const double *cl = (corr && !corr_fused) ? s_qn->d : nullptr;
  const double *cu = (corr && !corr_fused) ? y_qn->d : nullptr;
  if (corr && corrector_fused && !corr_fused) {
    set_error("internal: fused corrector solve with a panel of %d columns", m);
    return PO_ERR_ARG;
  }
  // t = Dinv o d1 and P^T t were produced by setUpKKTSystem's Gram pass when the right-hand side was known then
  const bool have_t0 = !refine_pass && t0_valid && t0_mu == mu && !corr && (int)t0dots.size() == m;
  std::vector<double> dots(m > 0 ? m : 1, 0.0);
  if (corr_fused) {
    // corrector products, t = Dinv o d1 and P^T t in ONE pass (the bits of k_corrector + k_d1 + k_mdot)
    need_panel();
    PO_TRY(k_corr_d1_dots(ctx, bounds(), px->d, pzl->d, pzu->d, rx->d, Dinv->d, beta_mu, P.data(), m, n, tvec->d,
                          dots.data()));
  } else {
    if (!refine_pass && !have_t0) PO_TRY(k_d1(ctx, bounds(), rx->d, Dinv->d, beta_mu, n, tvec->d, cl, cu));
    if (have_t0) {
      for (int i = 0; i < m; i++) dots[i] = t0dots[i];
    } else if (refine_pass && tdots_valid && (int)tdots.size() == m) {
      dots = tdots;  // P^T t' came out of the fused first pass (k_solve2_dots)
    } else if (m > 0) {
      need_panel();
      PO_TRY(k_mdot(ctx, tvec->d, P.data(), m, n, dots.data()));
    }
  }
  if (!refine_pass) first_t_recomputable = have_t0 && t_is_plain_dinv_d1;  // tvec / Dinv are dinv_d1's for this mu
  if (!refine_pass) t0_valid = false;  // tvec is overwritten by the passes below
  tdots_valid = false;
  // yz = G^-1 (d3 - A yx0)   (:2150-2159)
  std::vector<double> yz(c > 0 ? c : 1, 0.0), yz2(c > 0 ? c : 1, 0.0), zeta(k > 0 ? k : 1, 0.0);
  for (int i = 0; i < c; i++) {
    yz[i] = (b.z[i] + (b.zs[i] + vars.s[i] * b.s[i]) / vars.zs[i] -
             (b.zt[i] + vars.t[i] * b.t[i]) / vars.zt[i] - dots[i]);
  }
  if (c > 0) lu_solve(c, Gf.data(), c, gpiv.data(), yz.data());
  if (k > 0) {
    // Z^T px0 = Z^T Dinv d1 + W_ZA yz ; zeta = Ce^-1 (Z^T px0) ; yz2 = G^-1 (-W_AZ zeta)
    for (int i = 0; i < k; i++) {
      double v = dots[c + i];
      for (int l = 0; l < c; l++) v += W[(c + i) + (size_t)m * l] * yz[l];
      zeta[i] = v;
    }
    lu_solve(k, Cef.data(), k, cpiv.data(), zeta.data());
    for (int i = 0; i < c; i++) {
      double v = 0.0;
      for (int j = 0; j < k; j++) v += W[i + (size_t)m * (c + j)] * zeta[j];
      yz2[i] = -v;
    }
    if (c > 0) lu_solve(c, Gf.data(), c, gpiv.data(), yz2.data());
  }
  std::vector<double> alpha(m > 0 ? m : 1, 0.0);
  for (int i = 0; i < c; i++) alpha[i] = yz[i] - yz2[i];
  for (int j = 0; j < k; j++) alpha[c + j] = -zeta[j];
  // P^T (t + Dinv P alpha) = dots + W alpha
  if (!refine_pass) ptpx.assign(m > 0 ? m : 1, 0.0);
  for (int i = 0; i < m; i++) {
    double v = dots[i];
    for (int j = 0; j < m; j++) v += W[i + (size_t)m * j] * alpha[j];
    ptpx[i] = refine_pass ? ptpx[i] + v : v;
  }
  ptpx_valid = true;
  merit_cache_valid = false;  // the step is about to change
  px_amax_valid = false;
  fused_merit_valid = false;
  w_comp_valid = w_merit_cache_valid = false;
  pz_stored = true;
  // Fused refinement residual: the coefficients of addKKTResStep (:1475-1483) are known before
  // the axpy pass starts (A-part = p.z = alpha_A; Z-part = d0 M^-1 d0 Z^T px with Z^T px = ptpx),
  // so the same pass over P also emits the right-hand side t' of the refinement solve.
  const bool seq_lin = options.integer("sequential_linear_method");
  const int kq = (qn && !seq_lin) ? qn->size() : 0;
  // (a panel wider than one kernel's argument tables takes the plain sequence: the launchers collapse it, kernels.hip)
  const bool fuse = fuse_residual && !refine_pass && analytic_panel_dots && kq == k &&
                    !options.integer("use_diag_hessian") && !inexact_newton_step && m <= kMaxPanel;
  vA_valid = true;
  std::vector<double> coef(m > 0 ? m : 1, 0.0);
  double diag = options.real("qn_sigma");
  if (fuse) {
    for (int i = 0; i < c; i++) coef[i] = alpha[i];
    if (qn && !seq_lin) {
      diag += qn->diag();
      if (k > 0) {
        std::vector<double> rz(ptpx.begin() + c, ptpx.begin() + c + k);
        qn->applyCompactInverse(rz.data());
        for (int j = 0; j < k; j++) coef[c + j] = rz[j];
      }
    }
  }
  // Unformed L-SR1 columns (setUpKKTSystem left them unformed on purpose): the two fused passes take the panel as
  // [Y_0..Y_{k-1} | Ac] with the S partners beside it; coefficients and products are permuted on the host.
  std::vector<const double *> Yp, Sp;
  std::vector<double *> Zo;
  double b0z = 0.0;
  const bool first_fused = fuse && fused_dots && m > 0 && !corr;
  const bool virt = virtual_z && recompute_first_step && recompute_rhs && !have_panel && k > 0 && k <= kMaxVirt &&
                    (first_fused || (refine_pass && step_deferred && virt_first)) && qn &&
                    qn->pendingZ(&Yp, &Sp, &Zo, &b0z) && (int)Yp.size() == k;
  std::vector<const double *> Pv;
  auto to_virt = [&](const std::vector<double> &a) {  // [Ac | Z] order -> [Z | Ac] order
    std::vector<double> r(m > 0 ? m : 1, 0.0);
    for (int j = 0; j < k; j++) r[j] = a[c + j];
    for (int i = 0; i < c; i++) r[k + i] = a[i];
    return r;
  };
  if (virt) {
    Pv = Yp;
    for (Vec *a : Ac) Pv.push_back(a->d);
  } else {
    need_panel();
  }
  if (first_fused) {
    // one pass: t' = refinement rhs and P^T t' for the refinement solve.  The step itself (px, pzl, pzu, A^T pz) is
    // NOT stored: the refinement pass recomputes it from (t, alpha) in registers (k_solve2r) -- four output streams
    // less, and an HBM write costs about four reads here.  t' goes to xt (free during the solves), t stays in tvec.
    std::vector<double> out(m + 2, 0.0);
    const bool defer = recompute_first_step;
    // ... and with recompute_rhs not even t': the pass only takes the products P^T t', the refinement pass forms t'
    // again from the residual coefficients (an output stream costs about four input streams)
    double *tp_out = !defer ? tvec->d : (recompute_rhs ? nullptr : xt->d);
    // Dinv and t re-formed in the element epilogue from the bound data and rx it loads anyway (same bits)
    const bool redo_dt1 = recompute_dt && defer && recompute_rhs && first_t_recomputable &&
                          dbg_switch(SW_REDO_DT1, nullptr, 1) != 0;
    const double *t_in = redo_dt1 ? nullptr : tvec->d;
    if (virt) {
      const std::vector<double> av = to_virt(alpha), cv = to_virt(coef);
      PO_TRY(k_solve2_dots(ctx, bounds(), t_in, Dinv->d, av.data(), cv.data(), Pv.data(), m, beta_mu, tau, rx->d,
                           diag, n, px->d, pzl->d, pzu->d, tp_out, vA->d, c, out.data(), nullptr, 0, k, Sp.data(), k,
                           b0z, t0_diag));
      tdots.assign(m, 0.0);
      for (int j = 0; j < k; j++) tdots[c + j] = out[j];
      for (int i = 0; i < c; i++) tdots[i] = out[k + i];
    } else {
      PO_TRY(k_solve2_dots(ctx, bounds(), t_in, Dinv->d, alpha.data(), coef.data(), P.data(), m,
                           beta_mu, tau, rx->d, diag, n, px->d, pzl->d, pzu->d, tp_out, vA->d, c,
                           out.data(), nullptr, defer ? 0 : 1, 0, nullptr, 0, 0.0, t0_diag));
      tdots.assign(out.begin(), out.begin() + m);
    }
    virt_first = virt;
    tdots_valid = true;
    step_mins[0] = out[m];
    step_mins[1] = out[m + 1];
    step_deferred = defer;
    if (defer) {
      alpha_first = alpha;
      coef_first = coef;
      diag_first = diag;
    }
  } else if (refine_pass && step_deferred) {
    step_deferred = false;
    // the same sweep takes the sums the complementarity check of scaleKKTStep and the merit derivative need of the
    // final step (see solve2r_kernel): no separate pass over the step afterwards
    const bool take_merit = fuse_merit && recompute_rhs && !corr && dbg_switch(SW_FUSED_MERIT, nullptr, 1) != 0;
    const double *gm = take_merit ? g->d : nullptr;
    double *mo = take_merit ? fused_merit : nullptr;
    // lean step: (pzl, pzu) stay in registers; their only consumer left, the multiplier update, re-forms them
    const bool lean = lean_step && lean_step_allowed && take_merit && iterate_logs_valid &&
                      dbg_switch(SW_LEAN_STEP, nullptr, 1) != 0;
    double *pzl_out = lean ? nullptr : pzl->d, *pzu_out = lean ? nullptr : pzu->d;
    // Dinv and the first right-hand side t re-formed in registers from data the pass loads anyway (same bits)
    const bool redo_dt = recompute_dt && recompute_rhs && first_t_recomputable &&
                         dbg_switch(SW_REDO_DT, nullptr, 1) != 0;
    const double *t1p = redo_dt ? nullptr : tvec->d;
    if (lean) {
      pz_stored = false;
      step_beta_mu = beta_mu;
    }
    if (virt) {
      const std::vector<double> a1v = to_virt(alpha_first), a2v = to_virt(alpha), crv = to_virt(coef_first);
      PO_TRY(k_solve2r(ctx, bounds(), t1p, nullptr, Dinv->d, a1v.data(), a2v.data(), Pv.data(), m, beta_mu, tau, n,
                       px->d, pzl_out, pzu_out, vA->d, c, step_mins, crv.data(), rx->d, diag_first, k, Sp.data(), k,
                       b0z, gm, mo, t0_diag));
    } else {
      PO_TRY(k_solve2r(ctx, bounds(), t1p, recompute_rhs ? nullptr : xt->d, Dinv->d, alpha_first.data(),
                       alpha.data(), P.data(), m, beta_mu, tau, n, px->d, pzl_out, pzu_out, vA->d, c, step_mins,
                       coef_first.data(), rx->d, diag_first, 0, nullptr, 0, 0.0, gm, mo, t0_diag));
    }
    if (take_merit) {
      after_reduce(ctx, [this] {
        step_mins[0] = fused_merit[7];
        step_mins[1] = fused_merit[8];
        fused_merit_valid = true;
      });
    }
  } else if (corr_fused) {
    step_deferred = false;
    // corrector solve: the corrector terms re-formed from the affine step this pass overwrites, and the sums of
    // scaleKKTStep / evalMeritInitDeriv of the step it has in registers (k_comp_merit and its round trip disappear)
    const bool want_logs = !iterate_logs_valid;
    PO_TRY(k_solve2c(ctx, bounds(), tvec->d, Dinv->d, alpha.data(), P.data(), m, beta_mu, tau, n, px->d, pzl->d,
                     pzu->d, vA->d, c, g->d, want_logs ? 1 : 0, corr_out));
    after_reduce(ctx, [this, want_logs] {
      for (int i = 0; i < 7; i++) fused_merit[i] = corr_out[i];
      fused_merit[7] = step_mins[0] = corr_out[9];
      fused_merit[8] = step_mins[1] = corr_out[10];
      fused_merit[9] = corr_out[11];
      if (want_logs) {  // the barrier sums of the iterate, as k_comp_merit takes them
        iterate_logs[0] = corr_out[7];
        iterate_logs[1] = corr_out[8];
        iterate_logs_valid = true;
      }
      fused_merit_valid = true;
    });
  } else {
    if (!refine_pass) step_deferred = false;
    PO_TRY(k_solve2(ctx, bounds(), tvec->d, Dinv->d, alpha.data(), P.data(), m, beta_mu,
                    refine_pass ? 1 : 0, tau, n, px->d, pzl->d, pzu->d, step_mins,
                    fuse ? coef.data() : nullptr, rx->d, diag, tvec->d, vA->d, c, cl, cu));
  }
  residual_fused = fuse;
  // dense blocks: full solve (:2165-2170) minus the bx-only solve (:2300-2305)
  for (int i = 0; i < c; i++) {
    const double zs1 = yz[i] - b.s[i];
    const double zt1 = -b.t[i] - yz[i];
    out.z[i] = yz[i] - yz2[i];
    out.zs[i] = zs1 - yz2[i];
    out.zt[i] = zt1 + yz2[i];
    out.s[i] = (b.zs[i] - vars.s[i] * zs1) / vars.zs[i] + (vars.s[i] * yz2[i]) / vars.zs[i];
    out.t[i] = (b.zt[i] - vars.t[i] * zt1) / vars.zt[i] - (vars.t[i] * yz2[i]) / vars.zt[i];
  }
  return PO_OK;
}

int InteriorPoint::computeKKTStepWithRefinement(double mu, bool use_qn, double tau) {
  if (has_w) return computeKKTStepWithRefinementW(mu, use_qn, tau);
  const int nref = options.integer("iterative_refinement_steps");
  const double beta_mu = options.real("rel_bound_barrier") * mu;
  denseResidual(mu, res);
  PO_TRY(solveKKT(res, mu, use_qn, false, tau, step, nref > 0));
  for (int it = 0; it < nref; it++) {  // :4985-4991
    // dots of the current step with [Ac | Z_qn]: A px for r'.z, Z^T px for B px
    const bool with_qn = qn && !options.integer("sequential_linear_method");
    int kq = with_qn ? qn->size() : 0;
    // (the panel is only asked for when a pass below streams it: zPointers() forms unformed L-SR1 columns)
    std::vector<const double *> Pq;
    bool have_pq = false;
    auto need_pq = [&]() {
      if (!have_pq) {
        int k2 = 0;
        Pq = panel(with_qn, &k2);
        have_pq = true;
      }
    };
    const int mq = c + kq;
    std::vector<double> dots(mq > 0 ? mq : 1, 0.0);
    if (analytic_panel_dots && ptpx_valid && mq == c + wk) {
      for (int i = 0; i < mq; i++) dots[i] = ptpx[i];
    } else if (mq > 0) {
      need_pq();
      PO_TRY(k_mdot(ctx, px->d, Pq.data(), mq, n, dots.data()));
    }
    double diag = options.real("qn_sigma");
    std::vector<double> coef(mq + 1, 0.0);
    for (int i = 0; i < c; i++) coef[i] = step.z[i];
    int mres = mq;
    if (inexact_newton_step || options.integer("use_diag_hessian")) {
      // addKKTResStep :1461-1473: -H px (Hessian-vector product) or -h o px replaces the whole
      // quasi-Newton term, sigma included; it rides as one more panel column with coefficient -1
      diag = 0.0;
      if (inexact_newton_step) {
        if (prob->evalHvecProduct(x, vars.z.data(), nullptr, px, xt) != 0) return PO_ERR_USER;
      } else {
        PO_TRY(k_mul(ctx, xt->d, 1.0, hdiag->d, px->d, n));
      }
      need_pq();
      Pq.push_back(xt->d);
      coef[mq] = -1.0;
      mres = mq + 1;
    } else if (qn && !options.integer("sequential_linear_method")) {
      diag += qn->diag();
      if (kq > 0) {
        std::vector<double> rz(dots.begin() + c, dots.begin() + c + kq);
        qn->applyCompactInverse(rz.data());
        for (int j = 0; j < kq; j++) coef[c + j] = rz[j];
      }
    }
    if (!(it == 0 && residual_fused)) {
      need_pq();
      PO_TRY(k_res_step(ctx, bounds(), rx->d, px->d, pzl->d, pzu->d, Dinv->d, coef.data(), Pq.data(),
                        mres, diag, beta_mu, n, tvec->d));
    }
    Dense r2;
    r2.resize(c);
    denseResidual(mu, r2);
    for (int i = 0; i < c; i++) {  // addKKTResStep dense rows :1529-1535
      r2.z[i] -= (dots[i] - step.s[i] + step.t[i]);
      r2.s[i] += (step.zs[i] - step.z[i]);
      r2.t[i] += (step.zt[i] + step.z[i]);
      r2.zs[i] -= (step.s[i] * vars.zs[i] + vars.s[i] * step.zs[i]);
      r2.zt[i] -= (step.t[i] * vars.zt[i] + vars.t[i] * step.zt[i]);
    }
    PO_TRY(solveKKT(r2, mu, use_qn, true, tau, refine));
    for (int i = 0; i < c; i++) {
      step.z[i] += refine.z[i];
      step.s[i] += refine.s[i];
      step.t[i] += refine.t[i];
      step.zs[i] += refine.zs[i];
      step.zt[i] += refine.zt[i];
    }
  }
  sx = sz = 1.0;
  return PO_OK;
}

int InteriorPoint::checkKKTStep(int iteration, double mu) {
  const double beta_mu = options.real("rel_bound_barrier") * mu;
  if (!pz_stored) {  // lean step: materialise the bound-multiplier steps exactly as the update will form them
    PO_TRY(k_form_pz(ctx, bounds(), px->d, step_beta_mu, n, pzl->d, pzu->d));
    pz_stored = true;
  }
  const bool seq_lin = options.integer("sequential_linear_method");
  int kq = 0;
  std::vector<const double *> Pq = panel(qn && !seq_lin, &kq);
  const int mq = c + kq;
  std::vector<double> dots(mq > 0 ? mq : 1, 0.0), coef(mq > 0 ? mq : 1, 0.0);
  if (mq > 0) PO_TRY(k_mdot(ctx, px->d, Pq.data(), mq, n, dots.data()));  // explicit: this is a check
  for (int i = 0; i < c; i++) coef[i] = step.z[i];
  double diag = options.real("qn_sigma");
  if (qn && !seq_lin && !options.integer("use_diag_hessian")) {
    diag += qn->diag();
    if (kq > 0) {
      std::vector<double> rz(dots.begin() + c, dots.begin() + c + kq);
      qn->applyCompactInverse(rz.data());
      for (int j = 0; j < kq; j++) coef[c + j] = rz[j];
    }
  }
  if (has_w) {  // the sparse multiplier step enters r'x through Aw^T pzw: one more column
    if (prob->setSparseJacobianTranspose(1.0, x, wstepv[0], xt) != 0) return PO_ERR_USER;
    Pq.push_back(xt->d);
    coef.push_back(1.0);
  }
  double mx[3];
  PO_TRY(computeResidual(mu, true));  // fresh rx, as the reference recomputes it (:6216)
  PO_TRY(k_step_check(ctx, bounds(), rx->d, px->d, pzl->d, pzu->d, coef.data(), Pq.data(), (int)Pq.size(), diag,
                      beta_mu, n, mx));
  Dense r;
  r.resize(c);
  denseResidual(mu, r);
  double mz = 0.0, ms = 0.0, mt = 0.0, mzs = 0.0, mzt = 0.0;
  for (int i = 0; i < c; i++) {  // addKKTResStep dense rows :1529-1535
    mz = std::max(mz, fabs(r.z[i] - (dots[i] - step.s[i] + step.t[i])));
    ms = std::max(ms, fabs(r.s[i] + (step.zs[i] - step.z[i])));
    mt = std::max(mt, fabs(r.t[i] + (step.zt[i] + step.z[i])));
    mzs = std::max(mzs, fabs(r.zs[i] - (step.s[i] * vars.zs[i] + vars.s[i] * step.zs[i])));
    mzt = std::max(mzt, fabs(r.zt[i] - (step.t[i] * vars.zt[i] + vars.t[i] * step.zt[i])));
  }
  if (ctx->rank == 0) {
    char line[640];
    snprintf(line, sizeof(line),
             "\nResidual step check for iteration %d:\n"
             "max |stationarity (x block)|:             %10.4e\n"
             "max |dense constraints (z block)|:        %10.4e\n"
             "max |slack duals (s block)|:              %10.4e\n"
             "max |slack duals (t block)|:              %10.4e\n"
             "max |slack complementarity (zs block)|:   %10.4e\n"
             "max |slack complementarity (zt block)|:   %10.4e\n"
             "max |lower-bound complementarity (zl)|:   %10.4e\n"
             "max |upper-bound complementarity (zu)|:   %10.4e\n",
             iteration, mx[0], mz, ms, mt, mzs, mzt, mx[1], mx[2]);
    history += line;
  }
  return PO_OK;
}

int InteriorPoint::checkGradients(double dh, std::string *report) {  // :6196-6199
  return prob->checkGradients(dh, x, options.integer("use_hvec_product"), xt, tvec, report);
}

int InteriorPoint::checkMeritFuncGradient(Vec *xpt, double dh, double out[2]) {  // :3280-3432
  if (xpt) PO_TRY(k_copy(ctx, x->d, xpt->d, n));
  cwx_valid = false;
  if (prob->evalObjCon(x, &fobj, cvals.data()) != 0) {
    fprintf(stderr, "ParOpt: Function and constraint evaluation failed\n");
    return PO_ERR_USER;
  }
  neval++;
  if (prob->evalObjConGradient(x, g, Ac.data()) != 0) {
    fprintf(stderr, "ParOpt: Gradient evaluation failed\n");
    return PO_ERR_USER;
  }
  ngeval++;
  ac_valid = true;
  acz_valid = false;
  if (!pz_stored) {  // a lean step keeps px only; nothing below reads pzl / pzu
    pz_stored = true;
  }
  if (xpt) {  // a direction of our own: px = -g / |g|, fixed slack steps, zero elsewhere (:3327-3352)
    double g2 = 0.0;
    PO_TRY(k_reduce1(ctx, RED_SUMSQ, g->d, nullptr, n, &g2));
    PO_TRY(k_panel_axpy(ctx, px->d, g2 > 0.0 ? -1.0 / sqrt(g2) : 0.0, g->d, 0.0, nullptr, nullptr, 0, n));
    for (int i = 0; i < c; i++) {
      step.s[i] = -0.259 * (1 + (i % 3));
      step.t[i] = -0.349 * (4 - (i % 2));
    }
    if (has_w && nw > 0) {
      std::vector<double> hs((size_t)nw), ht((size_t)nw);
      for (int64_t i = 0; i < nw; i++) {
        hs[i] = -0.419 * (1 + (i % 5));
        ht[i] = -0.7513 * (1 + (i % 19));
      }
      PO_HIP(hipMemcpyAsync(wstepv[1]->d, hs.data(), sizeof(double) * (size_t)nw, hipMemcpyHostToDevice, ctx->stream));
      PO_HIP(hipMemcpyAsync(wstepv[2]->d, ht.data(), sizeof(double) * (size_t)nw, hipMemcpyHostToDevice, ctx->stream));
      PO_HIP(hipStreamSynchronize(ctx->stream));
    }
  }
  sx = sz = 1.0;
  ptpx_valid = false;
  merit_cache_valid = false;
  fused_merit_valid = false;
  w_comp_valid = w_merit_cache_valid = false;
  px_amax_valid = false;
  double m0 = 0.0, dm0 = 0.0;
  PO_TRY(evalMeritInitDeriv(1.0, &m0, &dm0));
  // the merit function at (x + dh px, s + dh ps, t + dh pt, sw + dh psw, tw + dh ptw) (:3386-3409)
  const double eps = 0.0;  // no clamping: the reference evaluates the perturbed point as it is
  double sums[2], wsums[5];
  std::vector<double> rs(c), rt(c), cs(cvals);
  for (int i = 0; i < c; i++) {
    rs[i] = vars.s[i] + dh * step.s[i];
    rt[i] = vars.t[i] + dh * step.t[i];
  }
  {
    // k_trial clamps into [lb + eps, ub - eps]; with eps = 0 an interior point is left alone
    PO_TRY(k_trial(ctx, bounds(), px->d, dh, eps, n, xt->d, sums));
    trial_logs_valid = false;
    s_qn_from_trial = false;
  }
  double ftemp = 0.0;
  if (prob->evalObjCon(xt, &ftemp, cs.data()) != 0) {
    fprintf(stderr, "ParOpt: Function and constraint evaluation failed\n");
    return PO_ERR_USER;
  }
  neval++;
  if (has_w) {
    if (prob->evalSparseCon(xt, wtmp) != 0) return PO_ERR_USER;
    PO_TRY(k_w_trial(ctx, wv(), wp(), dh, eps, gsw->d, gtw->d, wtmp->d, nw, wsums));
  }
  const double m1 = evalMeritFromSums(ftemp, cs.data(), rs.data(), rt.data(), sums[0], sums[1], has_w ? wsums : nullptr);
  const double fd = (m1 - m0) / dh;
  if (ctx->rank == 0) {
    fprintf(stdout, "Merit function test\n");
    fprintf(stdout, "dm FD: %15.8e  Actual: %15.8e  Err: %8.2e  Rel err: %8.2e\n", fd, dm0, fabs(fd - dm0),
            fabs((fd - dm0) / fd));
    fflush(stdout);
  }
  if (out) {
    out[0] = fd;
    out[1] = dm0;
  }
  return PO_OK;
}

int InteriorPoint::debugKKTStep(double mu) {
  PO_TRY(createQuasiNewton());
  PO_TRY(computeResidual(mu, true));
  PO_TRY(setUpKKTSystem(true));
  denseResidual(mu, res);
  if (has_w) {
    PO_TRY(solveKKTW(res, mu, true, false, 0.95, step));
  } else {
    PO_TRY(solveKKT(res, mu, true, false, 0.95, step));
  }
  sx = sz = 1.0;
  return PO_OK;
}

int InteriorPoint::debugSetState(const double *z, const double *s, const double *t, const double *zs,
                                 const double *zt, double mu) {
  PO_TRY(createQuasiNewton());
  for (int i = 0; i < c; i++) {
    vars.z[i] = z[i];
    vars.s[i] = s[i];
    vars.t[i] = t[i];
    vars.zs[i] = zs[i];
    vars.zt[i] = zt[i];
  }
  barrier_param = mu;
  {
    const std::string nt = options.str("norm_type");
    norm_type = nt == "infinity" ? 0 : (nt == "l1" ? 1 : 2);
  }
  // nothing carried between the passes of an iteration survives a state written from outside
  cwx_valid = trial_cw_valid = false;
  residual_cached = false;
  iterate_logs_valid = trial_logs_valid = fused_merit_valid = false;
  w_comp_valid = w_merit_cache_valid = merit_cache_valid = false;
  px_first_only = false;
  spec_enabled = spec_valid = false;
  acz_valid = false;
  ptpx_valid = tdots_valid = t0_valid = false;
  step_deferred = false;
  s_qn_from_trial = false;
  corrector_active = false;
  inexact_newton_step = false;
  pz_stored = true;
  panel_valid = false;
  if (prob->evalObjCon(x, &fobj, cvals.data()) != 0) return PO_ERR_USER;
  if (prob->evalObjConGradient(x, g, Ac.data()) != 0) return PO_ERR_USER;
  ac_valid = true;
  return PO_OK;
}

int InteriorPoint::debugKKT(double mu, int mode, double tau) {
  struct KeepSchur {  // G and Ce as assembled are copied for the dump of THIS call only
    bool &f;
    explicit KeepSchur(bool &f_) : f(f_) { f = true; }
    ~KeepSchur() { f = false; }
  } keep(debug_keep_schur);
  if (mode == 0) {
    PO_TRY(debugKKTStep(mu));
  } else if (mode == 2) {
    // the predictor-corrector step of optimize() from the injected state (round 6): residual and complementarity of the
    // iterate, KKT system for the affine right-hand side (mu = 0), then mehrotraStep -- affine solve + refinement, the
    // Mehrotra rule, corrector right-hand side and solve.  Leaves the new barrier parameter in barrier_param.
    PO_TRY(createQuasiNewton());
    const bool use_qn = !options.integer("sequential_linear_method");
    barrier_param = mu;
    PO_TRY(computeResidual(mu, true));
    const double comp = compFromSums(comp_prod, comp_count, vars, w_sums[0]);
    const double rhs_mu = 0.0;
    PO_TRY(setUpKKTSystem(use_qn, false, &rhs_mu));
    double tau_new = tau;
    PO_TRY(mehrotraStep(use_qn, comp, true, options.real("min_fraction_to_boundary"), options.real("abs_res_tol"),
                        &tau_new));
    (void)tau_new;
    PO_TRY(batch_flush(ctx));
  } else {
    PO_TRY(createQuasiNewton());
    PO_TRY(computeResidual(mu, true));
    allow_virtual_z = analytic_panel_dots && fused_dots && options.integer("iterative_refinement_steps") == 1 &&
                      !options.integer("use_diag_hessian") && !options.integer("sequential_linear_method");
    const int setup_rc = setUpKKTSystem(true, false, &mu);
    allow_virtual_z = false;
    PO_TRY(setup_rc);
    lean_step_allowed = false;  // (pzl, pzu) are stored: the test reads them
    PO_TRY(computeKKTStepWithRefinement(mu, true, tau));
    PO_TRY(batch_flush(ctx));
  }
  denseResidual(mu, res);
  resNorms(res, &debug_norms[0], &debug_norms[1], &debug_norms[2], &debug_norms[3]);
  return PO_OK;
}

// The Mehrotra strategies' step from a set-up KKT system (optimize() :4956-5045): affine (mu = 0) predictor with one
// refinement, probe to the boundary (tau = 1), complementarity there, sigma = (comp_affine / comp)^3 >= 0.01, the new
// barrier parameter, then either the corrector solve (predictor-corrector: affine products in the residual, no
// refinement) or a plain solve at the new parameter.  Leaves the step in (px, pzl, pzu, step), the new barrier
// parameter in barrier_param and the fraction to the boundary that goes with it in *tau_out.
int InteriorPoint::mehrotraStep(bool use_qn, double comp, bool corrector, double min_frac, double abs_res_tol,
                                double *tau_out) {
  // affine (mu = 0) predictor step, probed all the way to the boundary (:4956-5009)
  PO_TRY(computeKKTStepWithRefinement(0.0, use_qn, 1.0));
  double max_x = std::min(1.0, step_mins[0]), max_z = std::min(1.0, step_mins[1]);
  for (int i = 0; i < c; i++) {
    if (step.s[i] < 0.0) max_x = std::min(max_x, -vars.s[i] / step.s[i]);
    if (step.t[i] < 0.0) max_x = std::min(max_x, -vars.t[i] / step.t[i]);
    if (step.zs[i] < 0.0) max_z = std::min(max_z, -vars.zs[i] / step.zs[i]);
    if (step.zt[i] < 0.0) max_z = std::min(max_z, -vars.zt[i] / step.zt[i]);
  }
  double cs[2];
  if (fused_merit_valid && !has_w && dbg_switch(SW_MPC_POLY, "PAROPT_AMD_MPC_POLY", 1) != 0) {
    // the refinement pass of the affine solve took the complementarity polynomial of its step (solve2r_kernel):
    // S00 + ax S10 + az S01 + ax az S11 at the probe lengths, S00 / the bound count from the residual pass of this
    // iterate -- no pass over the step and no host round trip (round 6; scaleKKTStep uses the same form)
    cs[0] = comp_prod + max_x * fused_merit[0] + max_z * fused_merit[1] + max_x * max_z * fused_merit[2];
    cs[1] = comp_count;
  } else {
    PO_TRY(k_comp_step(ctx, bounds(), px->d, pzl->d, pzu->d, max_x, max_z, n, cs));
  }
  double prod = cs[0] / options.real("rel_bound_barrier"), count = cs[1];
  if (has_w) {
    double wprod = 0.0;
    PO_TRY(wCompStep(max_x, max_z, &wprod));
    prod += wprod;
    count += 2.0 * nw_global;
  }
  for (int i = 0; i < c; i++) {
    prod += ((vars.s[i] + max_x * step.s[i]) * (vars.zs[i] + max_z * step.zs[i]) +
             (vars.t[i] + max_x * step.t[i]) * (vars.zt[i] + max_z * step.zt[i]));
    count += 2.0;
  }
  const double comp_affine = count != 0.0 ? prod / count : 0.0;
  const double s1 = comp_affine / comp;
  double sigma = s1 * s1 * s1;
  if (sigma < 0.01) sigma = 0.01;
  barrier_param = sigma * comp;
  if (barrier_param < 0.09999 * abs_res_tol) barrier_param = 0.09999 * abs_res_tol;
  double tau = min_frac;
  if (1.0 - barrier_param >= tau) tau = 1.0 - barrier_param;
  *tau_out = tau;
  if (corrector) {
    // corrector: res.zl -= px*pzl, res.zu += px*pzu, res.zs -= ps*pzs, res.zt -= pt*pzt of the
    // affine step (:1729-1789); no refinement with the corrector (:5040-5041)
    // (one-pass corrector right-hand side + corrector solve with the merit sums: see solveKKT)
    corrector_fused = !has_w && c + wk >= 1 && c + wk <= kCorrDotsMax &&
                      dbg_switch(SW_MPC_FUSE, "PAROPT_AMD_MPC_FUSE", 1) != 0;
    if (!corrector_fused) PO_TRY(k_corrector(ctx, bounds(), px->d, pzl->d, pzu->d, n, s_qn->d, y_qn->d));
    denseResidual(barrier_param, res);
    for (int i = 0; i < c; i++) {
      res.zs[i] -= step.s[i] * step.zs[i];
      res.zt[i] -= step.t[i] * step.zt[i];
    }
    if (has_w) {
      PO_TRY(computeResidualW(barrier_param));
      PO_TRY(k_w_corrector(ctx, wp(), wr(), nw));
    }
    corrector_active = true;
    int rcs = has_w ? solveKKTW(res, barrier_param, use_qn, false, tau, step)
                    : solveKKT(res, barrier_param, use_qn, false, tau, step);
    corrector_active = false;
    corrector_fused = false;
    PO_TRY(rcs);
    sx = sz = 1.0;
  } else {
    PO_TRY(computeKKTStepWithRefinement(barrier_param, use_qn, tau));
  }
  return PO_OK;
}

// ================================================================================================
// step lengths
// ================================================================================================
int InteriorPoint::scaleKKTStep(double tau, double comp, double *alpha_x, double *alpha_z, int *ceq,
                                bool inexact) {  // :3196-3274 with computeMaxStep :2942-3103
  double ax = std::min(1.0, step_mins[0]), az = std::min(1.0, step_mins[1]);
  for (int i = 0; i < c; i++) {
    if (step.s[i] < 0.0) ax = std::min(ax, -tau * vars.s[i] / step.s[i]);
    if (step.t[i] < 0.0) ax = std::min(ax, -tau * vars.t[i] / step.t[i]);
    if (step.zs[i] < 0.0) az = std::min(az, -tau * vars.zs[i] / step.zs[i]);
    if (step.zt[i] < 0.0) az = std::min(az, -tau * vars.zt[i] / step.zt[i]);
  }
  *ceq = 0;
  if (inexact) {  // Newton step: one common step length (:3240-3248)
    if (ax > az) {
      ax = az;
    } else {
      az = ax;
    }
    sx = ax;
    sz = az;
    for (int i = 0; i < c; i++) {
      step.s[i] *= ax;
      step.t[i] *= ax;
      step.z[i] *= az;
      step.zs[i] *= az;
      step.zt[i] *= az;
    }
    *alpha_x = ax;
    *alpha_z = az;
    return PO_OK;
  }
  const double max_bnd = 100.0;
  if (ax > az) {
    if (ax > max_bnd * az) {
      ax = max_bnd * az;
    } else if (ax < az / max_bnd) {
      ax = az / max_bnd;
    }
  } else {
    if (az > max_bnd * ax) {
      az = max_bnd * ax;
    } else if (az < ax / max_bnd) {
      az = ax / max_bnd;
    }
  }
  double out[9], wprod = 0.0;
  if (fused_merit_valid && iterate_logs_valid && (!has_w || w_comp_valid)) {
    // (sparse constraints: the slacks' share of the polynomial came out of the step kernel, S00 = the complementarity
    // sum of the iterate's sparse slacks from its residual pass)
    if (has_w) wprod = w_sums[0] + ax * w_comp_poly[0] + az * w_comp_poly[1] + ax * az * w_comp_poly[2];
    // everything was taken by the refinement pass: the complementarity at the scaled step is
    // S00 + ax S10 + az S01 + ax az S11 with S00 / the bound count from the residual pass of this iterate.
    // Tolerance: near convergence S00 and the S10 / S01 terms nearly cancel, so the polynomial carries a relative
    // error of a few ulp of S00 where the direct sum of products has a few ulp of the result; the value only feeds the
    // comp_new > 10 comp test below (a factor-of-ten threshold), and the cmpEq tokens of all reference trajectories are
    // reproduced with it (tests/test_gpu_ip.py; A/B against the plain pass: test_write_saving_fusions_...).
    // Validity: comp_prod / iterate_logs belong to (x, zl, zu) as the last residual pass saw them.  Every internal
    // writer of the iterate clears the flags; an iteration callback or a getArray view must not WRITE the iterate
    // mid-solve (po_ip_set_iteration_callback: an observer) -- resetDesignAndBounds / readSolutionFile between solves
    // refresh mirrors and flags (refreshLiveMirror)
    out[0] = comp_prod + ax * fused_merit[0] + az * fused_merit[1] + ax * az * fused_merit[2];
    out[1] = comp_count;
    merit_cache[0] = iterate_logs[0];
    merit_cache[1] = iterate_logs[1];
    merit_cache[2] = fused_merit[3];
    merit_cache[3] = fused_merit[4];
    merit_cache[4] = fused_merit[5];
    merit_cache[5] = fused_merit[6];
    merit_cache[6] = fused_merit[9];
    merit_cache_valid = true;
  } else if (!has_w) {
    // one pass also yields the merit pieces and the step norm the line search is about to ask for
    PO_TRY(k_comp_merit(ctx, bounds(), px->d, pzl->d, pzu->d, ax, az, g->d, n, out));
    for (int i = 0; i < 7; i++) merit_cache[i] = out[2 + i];
    merit_cache_valid = true;
  } else {
    BatchScope batch(ctx);  // design and sparse parts of the complementarity: one collective + sync
    // (the design pass also yields the merit pieces and max|px| the line search is about to ask for, as on the
    // dense path: the separate merit pass and its max|px| reduction disappear)
    PO_TRY(k_comp_merit(ctx, bounds(), px->d, pzl->d, pzu->d, ax, az, g->d, n, out));
    PO_TRY(wCompStep(ax, az, &wprod));  // :2866-2889
    PO_TRY(batch.end());
    for (int i = 0; i < 7; i++) merit_cache[i] = out[2 + i];
    merit_cache_valid = true;
  }
  double prod = out[0] / options.real("rel_bound_barrier"), count = out[1];
  if (has_w) {
    prod += wprod;
    count += 2.0 * nw_global;
  }
  for (int i = 0; i < c; i++) {
    prod += ((vars.s[i] + ax * step.s[i]) * (vars.zs[i] + az * step.zs[i]) +
             (vars.t[i] + ax * step.t[i]) * (vars.zt[i] + az * step.zt[i]));
    count += 2.0;
  }
  const double comp_new = count != 0.0 ? prod / count : 0.0;
  if (comp_new > 10.0 * comp) {
    *ceq = 1;
    if (ax > az) {
      ax = az;
    } else {
      az = ax;
    }
  }
  // the n-sized parts of the step are scaled lazily; the dense parts right away
  sx = ax;
  sz = az;
  for (int i = 0; i < c; i++) {
    step.s[i] *= ax;
    step.t[i] *= ax;
    step.z[i] *= az;
    step.zs[i] *= az;
    step.zt[i] *= az;
  }
  *alpha_x = ax;
  *alpha_z = az;
  return PO_OK;
}

// ================================================================================================
// merit function and line search
// ================================================================================================
double InteriorPoint::evalMeritFromSums(double fk, const double *ck, const double *sk,
                                        const double *tk, double pos, double neg,
                                        const double *wsums) const {
  // evalMeritFunc :3569-3636 after the reduction of the bound terms; wsums (k_w_trial layout) =
  // {pos log, neg log, |cw - sw + tw|^2, gsw.sw, gtw.tw} of the sparse slacks (not scaled by beta)
  const double beta = options.real("rel_bound_barrier");
  pos *= beta;
  neg *= beta;
  if (wsums) {
    pos += wsums[0];
    neg += wsums[1];
  }
  for (int i = 0; i < c; i++) {
    if (sk[i] > 1.0) pos += log(sk[i]); else neg += log(sk[i]);
    if (tk[i] > 1.0) pos += log(tk[i]); else neg += log(tk[i]);
  }
  double dense_infeas = 0.0;
  for (int i = 0; i < c; i++) {
    const double cv = ck[i] - sk[i] + tk[i];
    dense_infeas += cv * cv;
  }
  double sparse_infeas = 0.0, wgam = 0.0;
  if (wsums) {
    sparse_infeas = sqrt(wsums[2]);
    wgam = wsums[3] + wsums[4];
  }
  const double infeas = sqrt(dense_infeas + sparse_infeas * sparse_infeas);
  double merit = (fk + wgam) - barrier_param * (pos + neg) + rho_penalty_search * infeas;
  for (int i = 0; i < c; i++) merit += gamma_s[i] * sk[i] + gamma_t[i] * tk[i];
  return merit;
}

int InteriorPoint::evalMeritInitDeriv(double max_x, double *merit_, double *pmerit_) {  // :3652-3924
  const double beta = options.real("rel_bound_barrier");
  const double abs_res_tol = options.real("abs_res_tol");
  const double frac = options.real("penalty_descent_fraction");
  const bool seq_lin = options.integer("sequential_linear_method");
  double out[6];
  double wm[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  const int kq = (qn && !seq_lin) ? qn->size() : 0;
  std::vector<const double *> Pq;  // (asked for only if streamed: zPointers() forms unformed L-SR1 columns)
  const int mq = c + kq;
  std::vector<double> dots(mq > 0 ? mq : 1, 0.0);
  {
    // design part, panel products (when not known analytically) and sparse part: one collective + sync (the
    // problem's sparse callbacks run in between: built-in problems only)
    BatchScope batch(ctx, has_w && prob->reductionsBatchable());
    if (merit_cache_valid) {
      out[0] = merit_cache[0];
      out[1] = merit_cache[1];
      out[2] = sx * merit_cache[2];
      out[3] = sx * merit_cache[3];
      out[4] = sx * merit_cache[4];
      out[5] = sx * sx * merit_cache[5];
    } else {
      PO_TRY(k_merit0(ctx, bounds(), px->d, sx, g->d, n, out));
      if (batch.open) {  // max|px| for the line search's minimum step rides in this batch (its own sync otherwise)
        PO_TRY(k_reduce1(ctx, RED_AMAX, px->d, nullptr, n, &px_amax_w));
        px_amax_valid = true;
      }
    }
    if (analytic_panel_dots && ptpx_valid && mq == c + wk) {
      for (int i = 0; i < mq; i++) dots[i] = ptpx[i];
    } else if (mq > 0) {
      int k2 = 0;
      Pq = panel(qn && !seq_lin, &k2);
      PO_TRY(k_mdot(ctx, px->d, Pq.data(), mq, n, dots.data()));
    }
    if (has_w && w_merit_cache_valid) {
      // taken at sx = 1 in the refinement batch (solveKKTW); every sum that depends on the step is linear in sx > 0
      for (int i = 0; i < 10; i++) wm[i] = w_merit_cache[i];
      wm[2] *= sx;
      wm[3] *= sx;
      wm[6] *= sx;
      wm[7] *= sx;
      wm[9] *= sx;
    } else if (has_w) {  // :3735-3765, 3489-3503
      const double *cw = nullptr;
      PO_TRY(sparseConAtIterate(&cw));
      PO_TRY(prob->setSparseJacobian(1.0, x, px, wtmp2));
      PO_TRY(k_w_merit(ctx, wv(), wp(), sx, gsw->d, gtw->d, cw, wtmp2->d, nw, wm));
    }
    PO_TRY(batch.end());
  }
  double pos = out[0] * beta, neg = out[1] * beta, ppos = out[2] * beta, pneg = out[3] * beta;
  const double gpx = out[4], pxpx = out[5];
  for (int i = 0; i < mq; i++) dots[i] *= sx;
  if (has_w) {
    pos += wm[0];
    neg += wm[1];
    ppos += wm[2];
    pneg += wm[3];
  }
  for (int i = 0; i < c; i++) {
    if (vars.s[i] > 1.0) pos += log(vars.s[i]); else neg += log(vars.s[i]);
    if (step.s[i] > 0.0) ppos += step.s[i] / vars.s[i]; else pneg += step.s[i] / vars.s[i];
    if (vars.t[i] > 1.0) pos += log(vars.t[i]); else neg += log(vars.t[i]);
    if (step.t[i] > 0.0) ppos += step.t[i] / vars.t[i]; else pneg += step.t[i] / vars.t[i];
  }
  // evalInfeasDeriv :3465-3509
  double dense_infeas = 0.0, pdense = 0.0;
  for (int i = 0; i < c; i++) {
    const double cval = cvals[i] - vars.s[i] + vars.t[i];
    const double pcval = dots[i] - step.s[i] + step.t[i];
    dense_infeas += cval * cval;
    pdense += cval * pcval;
  }
  const double sparse_infeas = has_w ? sqrt(wm[8]) : 0.0;
  const double psparse = wm[9];
  const double infeas = sqrt(dense_infeas + sparse_infeas * sparse_infeas);
  const double infeas_proj = infeas > 0.0 ? (pdense + psparse) / infeas : 0.0;
  // pTBp = 0.5 px^T B px  (:3820-3821), B px never formed
  double pTBp = 0.0;
  if (options.integer("use_diag_hessian") && hdiag) {  // sum px^2 h (no factor 1/2, :3810-3818)
    PO_TRY(k_mul(ctx, xt->d, 1.0, hdiag->d, px->d, n));
    PO_TRY(k_reduce1(ctx, RED_DOT, xt->d, px->d, n, &pTBp));
    pTBp *= sx * sx;
  } else if (qn && !seq_lin) {
    double v = qn->diag() * pxpx;
    if (kq > 0) {
      std::vector<double> rz(dots.begin() + c, dots.begin() + c + kq);
      std::vector<double> cf = rz;
      qn->applyCompactInverse(cf.data());
      for (int j = 0; j < kq; j++) v -= rz[j] * cf[j];
    }
    pTBp = 0.5 * v;
  }
  double merit = (fobj + (wm[4] + wm[5])) - barrier_param * (pos + neg);
  double pmerit = (gpx + (wm[6] + wm[7])) - barrier_param * (ppos + pneg);
  for (int i = 0; i < c; i++) {
    merit += gamma_s[i] * vars.s[i] + gamma_t[i] * vars.t[i];
    pmerit += gamma_s[i] * step.s[i] + gamma_t[i] * step.t[i];
  }
  double numer = pmerit;
  if (pTBp > 0.0) numer += 0.5 * pTBp;
  double rho_hat = 0.0;
  const bool small = infeas < 0.1 * abs_res_tol;
  if (small) {
    const double denom = -(1.0 - frac) * max_x * infeas;
    if (numer >= 0.0 && denom < 0.0) rho_hat = -(numer / denom);
  } else {
    double denom = infeas_proj + frac * max_x * infeas;
    if (numer >= 0.0) {
      if (denom < 0.0) {
        rho_hat = -(numer / denom);
      } else {
        denom = -(1.0 - frac) * max_x * infeas;
        rho_hat = -(numer / denom);
      }
    }
  }
  if (rho_hat > rho_penalty_search) {
    rho_penalty_search = rho_hat;
  } else {
    rho_penalty_search *= 0.5;
    if (rho_penalty_search < rho_hat) rho_penalty_search = rho_hat;
  }
  const double min_rho = options.real("min_rho_penalty_search");
  if (rho_penalty_search < min_rho) rho_penalty_search = min_rho;
  merit += rho_penalty_search * infeas;
  if (small) {
    pmerit -= rho_penalty_search * max_x * infeas;
  } else {
    pmerit += rho_penalty_search * infeas_proj;
  }
  *merit_ = merit;
  *pmerit_ = pmerit;
  return PO_OK;
}

static void clampStepDense(std::vector<double> &out, const std::vector<double> &v, double alpha,
                           const std::vector<double> &p, double eps, bool lower) {
  for (size_t i = 0; i < v.size(); i++) {
    double val = v[i] + alpha * p[i];
    if (lower && val <= 0.0 + eps) val = 0.0 + eps;
    out[i] = val;
  }
}

int InteriorPoint::lineSearch(double alpha_min, double *alpha_, double m0, double dm0,
                              int *fail_) {  // :3939-4156
  const int max_it = options.integer("max_line_iters");
  const bool backtrack = options.integer("use_backtracking_alpha");
  const double armijo = options.real("armijo_constant");
  const double fp = options.real("function_precision");
  const double eps = options.real("design_precision");
  double alpha = *alpha_;
  int fail = LS_FAILURE;
  const bool batchable = prob->reductionsBatchable();
  // the quasi-Newton step s = alpha sx px of the trial point is written by the trial pass itself (the accepted
  // trial is the last one): no separate pass in computeStepAndUpdate
  double *sq = (qn && options.integer("use_quasi_newton_update")) ? s_qn->d : nullptr;
  double merit = 0.0, best_merit = 0.0, best_alpha = -1.0;
  std::vector<double> rs(c), rt(c);
  int j = 0;
  for (; j < max_it; j++) {
    double sums[2], wsums[5];
    // the barrier sums at the trial point share the collective + host sync of the problem's own reductions
    // (f, c) when those go through the internal launchers (built-in problems)
    BatchScope batch(ctx, batchable);
    PO_TRY(k_trial(ctx, bounds(), px->d, alpha * sx, eps, n, xt->d, sums, sq));
    s_qn_a = alpha * sx;
    s_qn_from_trial = sq != nullptr;
    clampStepDense(rs, vars.s, alpha, step.s, eps, true);
    clampStepDense(rt, vars.t, alpha, step.t, eps, true);
    userBegin();
    int fail_obj = prob->evalObjCon(xt, &fobj, cvals.data());
    userEnd();
    trial_cw_valid = false;
    if (has_w && batchable && !fail_obj) {  // the sparse slack sums ride in the same collective
      if (prob->evalSparseCon(xt, wtmp) != 0) return PO_ERR_USER;
      trial_cw_valid = true;  // wtmp = cw(xt): handed to the iterate when this trial is accepted
      PO_TRY(k_w_trial(ctx, wv(), wp(), alpha * sx, eps, gsw->d, gtw->d, wtmp->d, nw, wsums));
    }
    PO_TRY(batch.end());
    trial_logs[0] = sums[0];  // barrier sums at the point xt holds now
    trial_logs[1] = sums[1];
    trial_logs_valid = true;
    neval++;
    if (fail_obj) {
      fprintf(stderr, "ParOpt: Evaluation failed during line search, trying new point\n");
      alpha *= 0.1;
      continue;
    }
    if (has_w && !batchable) {
      if (prob->evalSparseCon(xt, wtmp) != 0) return PO_ERR_USER;
      trial_cw_valid = true;
      PO_TRY(k_w_trial(ctx, wv(), wp(), alpha * sx, eps, gsw->d, gtw->d, wtmp->d, nw, wsums));
    }
    merit = evalMeritFromSums(fobj, cvals.data(), rs.data(), rt.data(), sums[0], sums[1],
                              has_w ? wsums : nullptr);
    if (best_alpha < 0.0 || merit < best_merit) {
      best_alpha = alpha;
      best_merit = merit;
    }
    if (merit - armijo * alpha * dm0 < (m0 + fp)) {
      fail = (fail & LS_MIN_STEP) ? (LS_SUCCESS | LS_MIN_STEP) : LS_SUCCESS;
      if ((merit <= m0 + fp) && (merit + fp >= m0)) fail |= LS_NO_IMPROVEMENT;
      break;
    } else if (fail & LS_MIN_STEP) {
      break;
    }
    if (j < max_it - 1) {
      if (backtrack) {
        alpha = 0.5 * alpha;
        if (alpha <= alpha_min) {
          alpha = alpha_min;
          fail |= LS_MIN_STEP;
        }
      } else {
        const double alpha_new = -0.5 * dm0 * (alpha * alpha) / (merit - m0 - dm0 * alpha);
        if (alpha_new <= alpha_min) {
          alpha = alpha_min;
          fail |= LS_MIN_STEP;
        } else if (alpha_new < 0.01 * alpha) {
          alpha = 0.01 * alpha;
        } else {
          alpha = alpha_new;
        }
      }
    }
  }
  if (j == max_it) fail |= LS_MAX_ITERS;
  if (!(fail & LS_SUCCESS)) {
    if (best_merit <= m0 + fp) {
      fail |= LS_SUCCESS;
      fail &= ~LS_FAILURE;
    } else if ((merit <= m0 + fp) && (merit + fp >= m0)) {
      fail |= LS_NO_IMPROVEMENT;
    }
    if (alpha != best_alpha) {
      alpha = best_alpha;
      trial_cw_valid = false;  // xt is rebuilt below without its sparse constraint values
      double sums[2];
      BatchScope batch(ctx, batchable);
      PO_TRY(k_trial(ctx, bounds(), px->d, alpha * sx, eps, n, xt->d, sums, sq));
      s_qn_a = alpha * sx;
      s_qn_from_trial = sq != nullptr;
      userBegin();
      int fail_obj = prob->evalObjCon(xt, &fobj, cvals.data());
      userEnd();
      PO_TRY(batch.end());
      trial_logs[0] = sums[0];
      trial_logs[1] = sums[1];
      trial_logs_valid = true;
      neval++;
      if (fail_obj) {
        fprintf(stderr, "ParOpt: Evaluation failed during line search\n");
        fail = LS_FAILURE;
      }
    } else {
      alpha = best_alpha;
    }
  }
  *alpha_ = alpha;
  *fail_ = fail;
  return PO_OK;
}

int InteriorPoint::computeStepAndUpdate(double alpha, int eval_obj_con, int perform_qn_update,
                                        int *update_type) {  // :4169-4267
  const bool use_qnu = options.integer("use_quasi_newton_update");
  const double eps = options.real("design_precision");
  *update_type = 0;
  const bool do_qn = qn && perform_qn_update && use_qnu;
  // The gradient difference y_qn = [-g + A^T z+]_old + [g - A^T z+]_new (:4197-4206, 4243-4251)
  // is assembled without streaming the constraint gradients: the first bracket from this
  // iteration's KKT residual rx = [lo]zl - [up]zu - g + A^T z and va = A^T pz (kept by the solves),
  // the second from the NEXT iteration's residual, which is evaluated right after the gradient and
  // reused at the top of the loop.
  // Sparse constraints (round 4): the same identity holds with A^T pz + Aw(x)^T pzw in place of A^T pz -- the first
  // bracket adds Aw(x_old)^T (zw+ - zw) = alpha sz Aw(x_old)^T pzw (:4207-4210 evaluate it at the OLD point), the second
  // is completed by the residual of the new point, whose panel carries Aw(x+)^T zw+ as one more column.  The sum is
  // formed here, from the final step, in one pass over the c constraint gradients: two panel passes, two sparse
  // transposes and the separate multiplier update disappear.  (Not with the linear-constraint recurrence, whose
  // `acz` follows the dense part of this vector alone.)
  const bool fast_w = has_w && do_qn && analytic_panel_dots && fast_yqn_w && !prob->linear_constraints;
  const bool fast_yqn = do_qn && analytic_panel_dots && ((!has_w && vA_valid) || fast_w);
  if (fast_w) {
    // vA <- sz Aw^T pzw + sum_j step.z[j] A_j   (step.z already carries sz: scaleKKTStep)
    std::vector<const double *> Aold;
    for (Vec *a : Ac) Aold.push_back(a->d);
    GroupCol gcol;  // (structured problems: the first term is formed by the pass itself)
    if (prob->sparseTransposeColumn(sz, x, wstepv[0], &gcol)) {
      PO_TRY(k_panel_axpy(ctx, vA->d, 0.0, nullptr, 1.0, step.z.data(), Aold.data(), c, n, &gcol));
    } else {
      if (prob->setSparseJacobianTranspose(sz, x, wstepv[0], vA) != 0) return PO_ERR_USER;
      if (c > 0) PO_TRY(k_panel_axpy(ctx, vA->d, 0.0, nullptr, 1.0, step.z.data(), Aold.data(), c, n));
    }
  }
  if (has_w) PO_TRY(k_w_update(ctx, wv(), wp(), alpha * sx, alpha * sz, eps, nw));  // :4177-4183
  // acz = A^T z follows z += alpha*sz*pz through va = A^T pz when the solves kept va; otherwise it is rebuilt
  const bool acz_follow = acz && acz_valid && vA_valid;
  // The bound-multiplier step and the first bracket of y_qn ride in the residual pass of the new point when that
  // pass follows anyway (kkt_res_update_kernel): nothing in between reads zl / zu.
  const bool fuse_upd = fast_yqn && fuse_mult_update;
  // No quasi-Newton update (a fixed approximation: the trust-region subproblem solves): the multiplier step still
  // rides in the residual pass of the new point, which is then taken here, right after the gradient, instead of at
  // the top of the next iteration (round 4: one pass over the bound data and one launch less per inner iteration)
  const bool fuse_upd_noqn = !do_qn && fuse_mult_update && !has_w && pz_stored;
  if (!pz_stored && !fuse_upd) {
    set_error("internal: lean step without the fused multiplier update");
    return PO_ERR_ARG;
  }
  MultUpdate upd;
  upd.a = upd.az = alpha * sz;
  if (fast_w) upd.az = alpha;  // vA was built from the scaled step
  upd.eps = eps;
  upd.acz_follow = acz_follow;
  if (fuse_upd || fuse_upd_noqn) {
    // (deferred)
  } else if (fast_yqn) {
    PO_TRY(k_update_mult_yqn(ctx, zl->d, pzl->d, zu->d, pzu->d, alpha * sz, eps, use_lower, use_upper,
                             rx->d, vA->d, upd.az, n, y_qn->d, acz_follow ? acz->d : nullptr));
  } else {
    PO_TRY(k_update_mult(ctx, zl->d, pzl->d, zu->d, pzu->d, alpha * sz, eps, use_lower, use_upper, n));
    if (acz_follow) PO_TRY(k_axpy(ctx, acz->d, alpha * sz, vA->d, n));
  }
  if (!acz_follow) acz_valid = false;
  for (int i = 0; i < c; i++) {
    double v = vars.s[i] + alpha * step.s[i];
    vars.s[i] = (v <= eps) ? eps : v;
    v = vars.t[i] + alpha * step.t[i];
    vars.t[i] = (v <= eps) ? eps : v;
    vars.z[i] = vars.z[i] + alpha * step.z[i];
    v = vars.zs[i] + alpha * step.zs[i];
    vars.zs[i] = (v <= eps) ? eps : v;
    v = vars.zt[i] + alpha * step.zt[i];
    vars.zt[i] = (v <= eps) ? eps : v;
  }
  std::vector<const double *> A;
  for (Vec *a : Ac) A.push_back(a->d);
  if (do_qn && !fast_yqn) {  // y_qn = -g + A^T z  at the old point with the new multipliers
    PO_TRY(k_panel_axpy(ctx, y_qn->d, -1.0, g->d, 0.0, vars.z.data(), A.data(), c, n));
    if (has_w && prob->addSparseJacobianTranspose(1.0, x, wvar[0], y_qn) != 0) return PO_ERR_USER;
  }
  if (eval_obj_con) {
    // the line search was skipped: form the new point now
    double sums[2];
    PO_TRY(k_trial(ctx, bounds(), px->d, alpha * sx, eps, n, xt->d, sums));  // (not in a batch: sums is local)
    s_qn_from_trial = false;
    trial_logs[0] = sums[0];
    trial_logs[1] = sums[1];
    trial_logs_valid = true;
  }
  // the accepted trial point IS the new design point (same clamp, same arithmetic)
  std::swap(x->d, xt->d);
  // ... and so are its sparse constraint values when the line search's last trial left them in wtmp
  cwx_valid = false;
  if (has_w && trial_cw_valid && !eval_obj_con) {
    std::swap(cwx->d, wtmp->d);
    cwx_valid = true;
  }
  trial_cw_valid = false;
  // ... and its barrier sums are those of the new iterate (trial_kernel and the merit pass take them alike)
  iterate_logs[0] = trial_logs[0];
  iterate_logs[1] = trial_logs[1];
  iterate_logs_valid = trial_logs_valid;
  trial_logs_valid = false;
  if (eval_obj_con) {
    userBegin();
    int fail = prob->evalObjCon(x, &fobj, cvals.data());
    userEnd();
    neval++;
    if (fail) {
      fprintf(stderr, "ParOpt: Function and constraint evaluation failed\n");
      return PO_ERR_USER;
    }
  }
  // a problem with linear dense constraints keeps the Jacobian of the first evaluation (Ac == nullptr)
  userBegin();
  // (constantJacobianMask: the library's own model problems may declare single columns constant; those are kept)
  std::vector<Vec *> Acp(Ac);
  const std::vector<char> *cmask = ac_valid ? prob->constantJacobianMask() : nullptr;
  for (int i = 0; cmask && i < c && i < (int)cmask->size(); i++)
    if ((*cmask)[i]) Acp[i] = nullptr;
  int fail_g = prob->evalObjConGradient(x, g, (prob->linear_constraints && ac_valid) ? nullptr : Acp.data());
  userEnd();
  ngeval++;
  if (fail_g) fprintf(stderr, "ParOpt: Gradient evaluation failed at final line search\n");
  if (do_qn) {
    if (!(s_qn_from_trial && s_qn_a == alpha * sx)) {
      PO_TRY(k_panel_axpy(ctx, s_qn->d, alpha * sx, px->d, 0.0, nullptr, nullptr, 0, n));
    }
    s_qn_from_trial = false;
    // the next residual's norms and the quasi-Newton products of the update are reduced together: nothing on the
    // host needs the norms before the update has its dots.  Only when no user code runs in between.
    // (user code between the two: only computeQuasiNewtonUpdateCorrection, and only when the problem has one)
    // (sparse constraints: y_qn is assembled by its own passes, the residual of the next iteration is still taken
    // here, early, so that its norms ride with the products of the update -- built-in problems only: their sparse
    // callbacks queue nothing that is read before the flush)
    const bool early_w = has_w && !fast_yqn && prob->reductionsBatchable();
    BatchScope batch(ctx, (fast_yqn || early_w) && qn->reductionsBatchable() &&
                              (prob->reductionsBatchable() || !prob->quasiNewtonCorrectionMayChangeStep()));
    if (fast_yqn) {
      // residual of the next iteration at (x+, z+, zl+, zu+): rx+ = [lo]zl+ - [up]zu+ - g+ + A+^T z+
      // ... and y_qn += [lo]zl+ - [up]zu+ - rx+ in the same pass (the residual kernel has all three in registers)
      // (sequential linear method: the quasi-Newton update that follows does not enter the next KKT diagonal, so this
      // pass can leave Dinv and t of the next first solve behind as the pass without an update does -- spec_dt_*)
      spec_dt_want = fuse_upd && options.integer("sequential_linear_method") != 0;
      const int rcr = computeResidual(barrier_param, true, y_qn, fuse_upd ? &upd : nullptr);
      spec_dt_want = false;
      PO_TRY(rcr);
      residual_cached = true;
    } else {
      std::vector<double> mz(c > 0 ? c : 1);
      for (int i = 0; i < c; i++) mz[i] = -vars.z[i];
      PO_TRY(k_panel_axpy(ctx, y_qn->d, 1.0, g->d, 1.0, mz.data(), A.data(), c, n));
      if (has_w && prob->addSparseJacobianTranspose(-1.0, x, wvar[0], y_qn) != 0) return PO_ERR_USER;
      if (early_w) {
        PO_TRY(computeResidual(barrier_param, true));
        residual_cached = true;
      }
    }
    int rcc = prob->computeQuasiNewtonUpdateCorrection(x, vars.z.data(), s_qn, y_qn);
    if (rcc != 0) return PO_ERR_USER;
    // Z^T s = alpha sx Z^T px is known from the solves (ptpx) when the step came from this very panel
    const int kz = qn->size();
    if (analytic_panel_dots && ptpx_valid && use_ztpx_hint && kz == wk && kz > 0 && (int)ptpx.size() == c + kz &&
        !inexact_newton_step && !prob->quasiNewtonCorrectionMayChangeStep()) {
      std::vector<double> zts(kz);
      for (int j = 0; j < kz; j++) zts[j] = alpha * sx * ptpx[c + j];
      qn->take_buffers = true;  // s_qn / y_qn are rewritten from scratch before their next use
      const int urc = qn->updateWithZTs(s_qn, y_qn, zts.data(), update_type);
      qn->take_buffers = false;
      PO_TRY(urc);
    } else {
      qn->take_buffers = true;
      const int urc = qn->update(s_qn, y_qn, update_type);
      qn->take_buffers = false;
      PO_TRY(urc);
    }
    PO_TRY(batch.end());
  } else if (qn && perform_qn_update) {  // :4261-4263
    if (qn->updateMult(x, vars.z.data(), has_w ? wvar[0] : nullptr) != 0) return PO_ERR_USER;
    *update_type = 0;
  }
  if (fuse_upd_noqn) {
    spec_dt_want = true;  // (the barrier parameter of the next first solve: spec_dt_mu, set by optimize())
    const int rc = computeResidual(barrier_param, true, nullptr, &upd);
    spec_dt_want = false;
    PO_TRY(rc);
    residual_cached = true;
  }
  return PO_OK;
}

// ================================================================================================
// optimize
// ================================================================================================
void InteriorPoint::getOptimizedPoint(Vec **x_, const double **z_, Vec **zl_, Vec **zu_) {
  if (x_) *x_ = x;
  if (z_) *z_ = vars.z.data();
  if (zl_) *zl_ = use_lower ? zl : nullptr;
  if (zu_) *zu_ = use_upper ? zu : nullptr;
}
void InteriorPoint::getOptimizedSlacks(const double **s_, const double **t_, const double **zs_,
                                       const double **zt_) {
  if (s_) *s_ = vars.s.data();
  if (t_) *t_ = vars.t.data();
  if (zs_) *zs_ = vars.zs.data();
  if (zt_) *zt_ = vars.zt.data();
}
void InteriorPoint::getIterationCounters(int *a, int *b, int *d) {
  if (a) *a = niter;
  if (b) *b = neval;
  if (d) *d = ngeval;
}

int InteriorPoint::optimize(const char *checkpoint) {
  // Host mirrors a caller obtained for the solver's vectors (getOptimizedPoint + getArray) are the vectors' data
  // while they are live (the reference's pointers ARE the data, src/ParOptVec.cpp:212-217): what the caller wrote
  // through them -- e.g. warm-start multipliers zl / zu for starting_point_strategy = no_start_strategy -- is
  // uploaded before the first kernel reads it, as po_vec_release_array(v, 1) would.  The mirrors stop being live
  // here (the kernels below do not look at mirrors): views handed out earlier go stale during the solve and are
  // refreshed by the next getArray.
  {
    Vec *mine[] = {x, zl, zu, wvar[0], wvar[1], wvar[2], wvar[3], wvar[4]};
    bool any = false;
    for (Vec *v : mine) {
      if (v && v->h_live && v->h && v->n > 0) {
        PO_HIP(hipMemcpyAsync(v->d, v->h, sizeof(double) * (size_t)v->n, hipMemcpyHostToDevice, ctx->stream));
        any = true;
      }
      if (v) v->h_live = 0;
    }
    if (any) PO_HIP(hipStreamSynchronize(ctx->stream));  // the pinned mirrors may be rewritten by the caller at once
  }
  PO_TRY(createQuasiNewton());
  const double abs_res_tol = options.real("abs_res_tol");
  const double rel_func_tol = options.real("rel_func_tol");
  const double fprec = options.real("function_precision");
  const double design_precision = options.real("design_precision");
  {
    const std::string nt = options.str("norm_type");
    norm_type = nt == "infinity" ? 0 : (nt == "l1" ? 1 : 2);
  }
  const std::string bname = options.str("barrier_strategy");
  enum { B_MONOTONE = 0, B_MEHROTRA = 1, B_MPC = 2, B_COMPFRAC = 3 };
  const int input_strategy = bname == "monotone" ? B_MONOTONE
                             : bname == "mehrotra" ? B_MEHROTRA
                             : bname == "mehrotra_predictor_corrector" ? B_MPC : B_COMPFRAC;
  int barrier_strategy = B_MONOTONE;  // always start monotone (:4427-4441)
  corrector_active = false;
  const bool use_hvec_product = options.integer("use_hvec_product");
  const bool use_diag_hessian = options.integer("use_diag_hessian");
  if (has_w && use_hvec_product) {
    for (int i = 0; i < 5; i++) {
      if (!wscalev[i]) wscalev[i] = vec_new(ctx, nw);
      if (!wscalev[i]) return PO_ERR_HIP;
    }
  }
  if (use_diag_hessian) PO_TRY(ensureHdiag());
  inexact_newton_step = false;
  barrier_param = options.real("init_barrier_param");
  rho_penalty_search = options.real("init_rho_penalty_search");
  const int max_major_iters = options.integer("max_major_iters");
  const bool use_qnu = options.integer("use_quasi_newton_update");
  const int hessian_reset_freq = options.integer("hessian_reset_freq");
  const bool seq_lin = options.integer("sequential_linear_method");
  const double min_frac = options.real("min_fraction_to_boundary");
  const bool use_line_search = options.integer("use_line_search");
  const int write_freq = options.integer("write_output_frequency");
  const int step_verification_frequency = options.integer("step_verification_frequency");
  const int gradient_verification_frequency = options.integer("gradient_verification_frequency");
  const std::string start = options.str("starting_point_strategy");
  niter = neval = ngeval = nhvec = 0;
  cwx_valid = trial_cw_valid = false;
  residual_cached = false;
  iterate_logs_valid = trial_logs_valid = fused_merit_valid = false;
  spec_dt_valid = spec_dt_want = false;
  w_comp_valid = w_merit_cache_valid = false;
  px_first_only = false;
  spec_enabled = spec_valid = false;
  history.clear();
  phase_names.clear();
  phase_seconds.clear();
  user_seconds = 0.0;
  user_pending = 0;
  ac_valid = false;
  acz_valid = false;
  if (!seq_lin && !qn && !use_diag_hessian) {
    if (ctx->rank == 0)
      fprintf(stderr,
              "ParOpt Error: Must use a sequential linear method if no quasi-Newton approximation "
              "is defined\n");
    return 1;
  }
  phaseBegin();
  PO_TRY(initAndCheckDesignAndBounds());
  userBegin();
  int fail_obj = prob->evalObjCon(x, &fobj, cvals.data());
  userEnd();
  neval++;
  if (fail_obj) {
    fprintf(stderr, "ParOpt: Initial function and constraint evaluation failed\n");
    return fail_obj;
  }
  userBegin();
  int fail_g = prob->evalObjConGradient(x, g, Ac.data());
  userEnd();
  ngeval++;
  if (fail_g) {
    fprintf(stderr, "ParOpt: Initial gradient evaluation failed\n");
    return fail_g;
  }
  ac_valid = true;
  if (start == "affine_step") {
    PO_TRY(initAffineStepMultipliers());
  } else if (start == "least_squares_multipliers") {
    PO_TRY(initLeastSquaresMultipliers());
  }
  if (qn && !use_qnu) {  // :4569-4573
    if (qn->updateMult(x, vars.z.data(), has_w ? wvar[0] : nullptr) != 0) return PO_ERR_USER;
  }
  phaseEnd("init");

  double fobj_prev = 0.0, alpha_prev = 0.0, alpha_xprev = 0.0, alpha_zprev = 0.0, dm0_prev = 0.0;
  double res_norm_prev = 0.0;
  int no_merit_function_improvement = 0, line_search_test = 0, line_search_failed = 0;
  char info[64];
  memset(info, 0, sizeof(info));
  char line[512];

  for (int k = 0; k < max_major_iters; k++, niter++) {
    int qn_hessian_reset = 0;
    if (qn && !seq_lin) {
      if (k > 0 && k % hessian_reset_freq == 0 && use_qnu) {
        qn->reset();
        qn_hessian_reset = 1;
      }
    }
    if (write_freq > 0 && k % write_freq == 0) {
      if (checkpoint) {
        if (writeSolutionFile(checkpoint) != PO_OK) checkpoint = nullptr;
      }
      prob->writeOutput(k, x);
    }
    if (iter_cb) iter_cb(iter_cb_user, k);
    // gradient_verification_frequency (:4522-4525, 4635-4639): finite-difference check of the user's gradients
    if (gradient_verification_frequency > 0 && (k % gradient_verification_frequency) == 0) {
      std::string rep;
      PO_TRY(prob->checkGradients(options.real("gradient_check_step_length"), x, use_hvec_product, xt, tvec, &rep));
      if (ctx->rank == 0) history += rep;
    }

    const bool rel_function_test =
        (alpha_xprev == 1.0 && alpha_zprev == 1.0 && (fabs(fobj - fobj_prev) < rel_func_tol * fabs(fobj_prev)));
    if (no_merit_function_improvement) {
      line_search_test += 1;
    } else {
      line_search_test = 0;
    }
    // complementarity + KKT residual + norms in ONE pass (:4656-4671)
    double max_prime = 0.0, max_dual = 0.0, max_infeas = 0.0, res_norm = 0.0;
    int monotone_barrier_converged = 0;
    double comp = 0.0;
    // (the inexact Newton step reads the 2-norms of the bound residuals, computeKKTGMRESStep: no shortcut there)
    spec_enabled = barrier_strategy == B_MONOTONE && norm_type == 0 && !has_w && !use_hvec_product;
    if (barrier_strategy == B_MONOTONE) {
      if (!residual_cached) PO_TRY(computeResidual(barrier_param, true));
      residual_cached = false;
      comp = compFromSums(comp_prod, comp_count, vars, w_sums[0]);
      denseResidual(barrier_param, res);
      resNorms(res, &max_prime, &max_dual, &max_infeas, &res_norm);
      if (k > 0 && ((res_norm < 10.0 * barrier_param) || rel_function_test || (line_search_test >= 2))) {
        monotone_barrier_converged = 1;
      }
      if (monotone_barrier_converged) {
        if (barrier_param > 0.1 * abs_res_tol) line_search_test = 0;
        const double mu_frac = options.real("monotone_barrier_fraction") * barrier_param;
        const double mu_pow = pow(barrier_param, options.real("monotone_barrier_power"));
        double new_mu = mu_frac;
        if (mu_pow < mu_frac) new_mu = mu_pow;
        if (new_mu < 0.1 * abs_res_tol) new_mu = 0.09999 * abs_res_tol;
        if (spec_valid && spec_mu == new_mu && norm_type == 0) {
          // the residual pass of this iterate took the two maxima for new_mu as well (the complementarity product
          // and the bound count do not depend on mu): what computeResidual(new_mu, false) would produce
          max_rzl = spec_max[0];
          max_rzu = spec_max[1];
          spec_valid = false;
        } else {
          PO_TRY(computeResidual(new_mu, false));  // rx does not depend on mu
        }
        denseResidual(new_mu, res);
        resNorms(res, &max_prime, &max_dual, &max_infeas, &res_norm);
        rho_penalty_search = options.real("min_rho_penalty_search");
        barrier_param = new_mu;
      }
    } else if (barrier_strategy == B_MEHROTRA || barrier_strategy == B_MPC) {  // :4737-4746
      if (!residual_cached) PO_TRY(computeResidual(barrier_param, true));
      residual_cached = false;
      comp = compFromSums(comp_prod, comp_count, vars, w_sums[0]);
      denseResidual(barrier_param, res);
      resNorms(res, &max_prime, &max_dual, &max_infeas, &res_norm);
    } else {  // complementarity fraction (:4747-4762)
      if (!residual_cached) PO_TRY(computeResidual(barrier_param, true));
      residual_cached = false;
      comp = compFromSums(comp_prod, comp_count, vars, w_sums[0]);
      barrier_param = options.real("monotone_barrier_fraction") * comp;
      if (barrier_param < 0.1 * abs_res_tol) barrier_param = 0.1 * abs_res_tol;
      PO_TRY(computeResidual(barrier_param, false));
      denseResidual(barrier_param, res);
      resNorms(res, &max_prime, &max_dual, &max_infeas, &res_norm);
    }
    phaseEnd("residual");

    if (ctx->rank == 0) {  // iteration table :4777-4801
      if (k == 0 || options.integer("output_level") > 0) {  // :4767-4774
        const char *inform = prob->sparseFactorInfo();
        if (inform) history += std::string("MatInfo: ") + inform + "\n";
      }
      if (k % 10 == 0) {
        snprintf(line, sizeof(line),
                 "\n%4s %4s %4s %4s %7s %7s %7s %12s %7s %7s %7s %7s %7s %8s %7s info\n", "iter",
                 "nobj", "ngrd", "nhvc", "alpha", "alphx", "alphz", "fobj", "|opt|", "|infes|",
                 "|dual|", "mu", "comp", "dmerit", "rho");
        history += line;
      }
      if (k == 0) {
        snprintf(line, sizeof(line),
                 "%4d %4d %4d %4d %7s %7s %7s %12.5e %7.1e %7.1e %7.1e %7.1e %7.1e %8s %7s %s\n", k,
                 neval, ngeval, nhvec, "--", "--", "--", fobj, max_prime, max_infeas, max_dual,
                 barrier_param, comp, "--", "--", info);
      } else {
        snprintf(line, sizeof(line),
                 "%4d %4d %4d %4d %7.1e %7.1e %7.1e %12.5e %7.1e %7.1e %7.1e %7.1e %7.1e %8.1e %7.1e %s\n",
                 k, neval, ngeval, nhvec, alpha_prev, alpha_xprev, alpha_zprev, fobj, max_prime,
                 max_infeas, max_dual, barrier_param, comp, dm0_prev, rho_penalty_search, info);
      }
      history += line;
    }

    int converged = 0;
    if (k > 0 && (barrier_param <= 0.1 * abs_res_tol) &&
        (res_norm < abs_res_tol || rel_function_test || (line_search_test >= 2))) {
      if (ctx->rank == 0) {
        if (rel_function_test) {
          history += "\nParOpt: Successfully converged on relative function test\n";
        } else if (line_search_test >= 2) {
          history +=
              "\nParOpt Warning: Current design point could not be improved. No barrier function "
              "decrease in previous two iterations\n";
        } else {
          history += "\nParOpt: Successfully converged to requested tolerance\n";
        }
      }
      converged = 1;
    }
    if (converged) {
      flushHistory();
      return 0;
    }

    const bool mehrotra = (barrier_strategy == B_MEHROTRA || barrier_strategy == B_MPC);
    // (what the first solve of the NEXT iteration will most likely take as its barrier parameter: see spec_dt_valid)
    spec_dt_mu = mehrotra ? 0.0 : barrier_param;
    double tau = min_frac;
    if (1.0 - barrier_param >= tau) tau = 1.0 - barrier_param;

    // inexact Newton-Krylov step with exact Hessian-vector products (:4853-4900)
    int gmres_iters = 0;
    inexact_newton_step = false;
    if (use_hvec_product) {
      const double gmres_rtol =
          res_norm_prev == 0.0 ? 1e300
                               : options.real("eisenstat_walker_gamma") *
                                     pow(res_norm / res_norm_prev, options.real("eisenstat_walker_alpha"));
      const double nk = options.real("nk_switch_tol");
      if (max_prime < nk && max_dual < nk && max_infeas < nk && gmres_rtol < options.real("max_gmres_rtol")) {
        const bool precon_qn = !(seq_lin || !options.integer("use_qn_gmres_precon"));
        PO_TRY(setUpKKTSystem(precon_qn));
        PO_TRY(computeKKTGMRESStep(gmres_rtol, options.real("gmres_atol"), precon_qn, tau, &gmres_iters));
        phaseEnd("gmres_step");
        // negative: GMRES did not produce a descent direction (:4883-4894), fall back to the quasi-Newton step
        if (gmres_iters > 0) inexact_newton_step = true;
      }
    }
    fobj_prev = fobj;
    res_norm_prev = res_norm;
    int seq_linear_step = 0, diagonal_quasi_newton_step = 0;
    bool use_qn = !seq_lin;
    if (inexact_newton_step) {
      // the step is already in place
    } else if (!seq_lin && line_search_failed && !use_qnu) {
      // a fixed quasi-Newton approximation whose line search failed (:4923-4939): sequential linear
      // step, or only the diagonal b0 of the approximation when that is positive
      use_qn = false;
      seq_linear_step = 1;
      if (qn && qn->diag() > 0.0) {
        seq_linear_step = 0;
        diagonal_quasi_newton_step = 1;
      }
    } else if (use_diag_hessian && !seq_lin) {  // :4940-4948
      // (an else-if chain in the reference, :4920-4949: under the sequential linear method the diagonal is never
      // evaluated -- hdiag keeps its initial zeros, which setUpKKTDiagSystem and addKKTResStep then use)
      use_qn = false;
      if (prob->evalHessianDiag(x, vars.z.data(), has_w ? wvar[0] : nullptr, hdiag) != 0) {
        fprintf(stderr, "ParOpt: Hessian diagonal evaluation failed\n");
        return PO_ERR_USER;
      }
    }

    if (inexact_newton_step) {
      // nothing to do: computeKKTGMRESStep left the step in (px, pzl, pzu, step)
    } else {
    const double rhs_mu = mehrotra ? 0.0 : barrier_param;  // of the solve that follows
    // unformed L-SR1 columns may stay unformed when every pass of this iteration that reads the panel is one of the
    // fused ones: a single quasi-Newton solve with one refinement step, no corrector solve
    allow_virtual_z = !mehrotra && use_qn && !diagonal_quasi_newton_step && analytic_panel_dots && fused_dots &&
                      options.integer("iterative_refinement_steps") == 1 && !options.integer("use_diag_hessian") &&
                      !options.integer("sequential_linear_method");
    const int setup_rc = setUpKKTSystem(use_qn, diagonal_quasi_newton_step != 0, &rhs_mu);
    allow_virtual_z = false;
    PO_TRY(setup_rc);
    phaseEnd("setup_kkt");
    if (!mehrotra) {
      // every consumer of (pzl, pzu) after this solve is the fused multiplier update of computeStepAndUpdate
      lean_step_allowed = qn && use_qnu && use_qn && !diagonal_quasi_newton_step && analytic_panel_dots &&
                          fuse_mult_update && !use_hvec_product && !use_diag_hessian && !seq_lin;
      const int step_rc = computeKKTStepWithRefinement(barrier_param, use_qn, tau);
      lean_step_allowed = false;
      PO_TRY(step_rc);
    } else {
      PO_TRY(mehrotraStep(use_qn, comp, barrier_strategy == B_MPC, min_frac, abs_res_tol, &tau));
    }
    }
    phaseEnd("kkt_step");
    // step_verification_frequency (:5056-5073): block maxima of the linearised KKT residual at the step
    if (step_verification_frequency > 0 && (k % step_verification_frequency) == 0 && !inexact_newton_step) {
      if (barrier_strategy == B_MPC && ctx->rank == 0)
        history += "Note: the step check is inconsistent with the predictor-corrector step (corrector terms)\n";
      PO_TRY(checkKKTStep(k, barrier_param));
    }

    double alpha_x = 1.0, alpha_z = 1.0;
    int ceq_step = 0;
    PO_TRY(scaleKKTStep(tau, comp, &alpha_x, &alpha_z, &ceq_step, inexact_newton_step));
    phaseEnd("scale_step");

    double alpha = 1.0;
    int line_fail = LS_FAILURE;
    int update_type = 0;
    int line_search_skipped = 0;
    no_merit_function_improvement = 0;

    if (use_line_search) {
      double m0 = 0.0, dm0 = 0.0;
      PO_TRY(evalMeritInitDeriv(alpha_x, &m0, &dm0));
      phaseEnd("merit_deriv");
      dm0_prev = dm0;
      if (dm0 >= 0.0 && dm0 <= fprec) {
        line_search_skipped = 1;
        PO_TRY(computeStepAndUpdate(alpha, 1, 1, &update_type));
        phaseEnd("step_update");
        if ((fobj_prev + fprec <= fobj) && (fobj + fprec <= fobj_prev)) line_fail = LS_NO_IMPROVEMENT;
      } else {
        if (dm0 >= 0.0) {  // :5130-5173
          if (qn) {
            qn_hessian_reset = 1;
            qn->reset();
          }
          diagonal_quasi_newton_step = 1;
          PO_TRY(computeResidual(barrier_param, true));
          denseResidual(barrier_param, res);
          resNorms(res, &max_prime, &max_dual, &max_infeas, &res_norm);
          PO_TRY(setUpKKTSystem(true, false, &barrier_param));
          PO_TRY(computeKKTStepWithRefinement(barrier_param, true, tau));
          PO_TRY(scaleKKTStep(tau, comp, &alpha_x, &alpha_z, &ceq_step));
          PO_TRY(evalMeritInitDeriv(alpha_x, &m0, &dm0));
          dm0_prev = dm0;
          phaseEnd("qn_reset_step");
        }
        if (dm0 >= 0.0) {
          line_fail = LS_FAILURE;
        } else {
          double px_norm = 0.0;
          if (merit_cache_valid) {
            px_norm = merit_cache[6];
          } else if (px_amax_valid) {
            px_norm = px_amax_w;
          } else {
            PO_TRY(k_reduce1(ctx, RED_AMAX, px->d, nullptr, n, &px_norm));
          }
          px_norm *= fabs(sx);
          double alpha_min = 1.0;
          if (px_norm != 0.0) alpha_min = fprec / px_norm;
          if (alpha_min > 0.5) alpha_min = 0.5;
          PO_TRY(lineSearch(alpha_min, &alpha, m0, dm0, &line_fail));
          phaseEnd("line_search");
          if (px_norm < design_precision) line_fail |= LS_SHORT_STEP;
          if (!(line_fail & LS_FAILURE)) {
            PO_TRY(computeStepAndUpdate(alpha, 0, 1, &update_type));
            phaseEnd("step_update");
          }
        }
      }
    } else {
      double m0 = 0.0, dm0 = 0.0;
      PO_TRY(evalMeritInitDeriv(alpha_x, &m0, &dm0));
      dm0_prev = dm0;
      line_fail = LS_SUCCESS;
      PO_TRY(computeStepAndUpdate(alpha, 1, 1, &update_type));
      // merit at the new point (:5236-5237): barrier sums of x itself (a zero step from x)
      double bs[2];
      PO_TRY(k_trial(ctx, bounds(), px->d, 0.0, design_precision, n, xt->d, bs));
      double wsums[5];
      if (has_w) {
        const double *cw = nullptr;
        PO_TRY(sparseConAtIterate(&cw));
        PO_TRY(k_w_trial(ctx, wv(), wp(), 0.0, design_precision, gsw->d, gtw->d, cw, nw, wsums));
      }
      const double m1 = evalMeritFromSums(fobj, cvals.data(), vars.s.data(), vars.t.data(), bs[0], bs[1],
                                          has_w ? wsums : nullptr);
      if ((m1 <= m0 + fprec) && (m1 + fprec >= m0)) {
        line_fail |= LS_NO_IMPROVEMENT;
      } else if (fabs(dm0) <= fprec) {
        line_fail = LS_NO_IMPROVEMENT;
      }
      phaseEnd("step_update");
    }

    no_merit_function_improvement =
        ((line_fail & LS_NO_IMPROVEMENT) || (line_fail & LS_MIN_STEP) ||
         (line_fail & LS_SHORT_STEP) || (line_fail & LS_FAILURE));
    line_search_failed = (line_fail & LS_FAILURE) ? 1 : 0;
    alpha_prev = alpha;
    alpha_xprev = alpha_x;
    alpha_zprev = alpha_z;
    if (qn && use_qnu && (line_fail & LS_FAILURE)) {
      qn_hessian_reset = 1;
      qn->reset();
    }
    {  // info tokens :5272-5322
      std::string s;
      if (gmres_iters != 0) s += "iNK" + std::to_string(gmres_iters) + " ";
      if (update_type == 1) s += "dampH ";
      else if (update_type == 2) s += "skipH ";
      if (qn_hessian_reset) s += "resetH ";
      if (line_fail & LS_FAILURE) s += "LFail ";
      if (line_fail & LS_MIN_STEP) s += "LMnStp ";
      if (line_fail & LS_MAX_ITERS) s += "LMxItr ";
      if (line_fail & LS_NO_IMPROVEMENT) s += "LNoImprv ";
      if (seq_linear_step) s += "SLP ";
      if (diagonal_quasi_newton_step) s += "DQN ";
      if (line_search_skipped) s += "LSkip ";
      if (ceq_step) s += "cmpEq ";
      memset(info, 0, sizeof(info));
      strncpy(info, s.c_str(), sizeof(info) - 1);
    }
    if (monotone_barrier_converged) barrier_strategy = input_strategy;
  }
  flushHistory();
  return 0;
}

// The reference streams the iteration table to `output_file` on the root rank (setOutputFile
// :1297-1309, :4777-4805); here it is accumulated in `history` and written when optimize returns.
void InteriorPoint::flushHistory() {
  userHarvest();
  phase_names.push_back("user_eval");
  phase_seconds.push_back(user_seconds);
  const std::string fname = options.str("output_file");
  // debugging aid: PAROPT_AMD_DUMP_LONG_SOLVES=<N> prints the head and tail of the iteration table of
  // every solve that took at least N major iterations (e.g. a stalled trust-region subproblem)
  if (ctx->rank == 0) {
    const char *dbg = getenv("PAROPT_AMD_DUMP_LONG_SOLVES");
    if (dbg && niter >= atoi(dbg)) {
      const size_t len = history.size();
      fprintf(stderr, "---- paropt_amd: solve with %d iterations ----\n%s\n   [...]\n%s\n", niter,
              history.substr(0, std::min<size_t>(len, 6000)).c_str(),
              history.substr(len > 4000 ? len - 4000 : 0).c_str());
    }
  }
  if (ctx->rank != 0 || fname.empty()) return;
  FILE *fp = fopen(fname.c_str(), "w");
  if (!fp) return;
  fputs("ParOptInteriorPoint (paropt_amd, MI355X)\n", fp);
  fputs(history.c_str(), fp);
  fclose(fp);
}

// writeSolutionFile (:883-972): ONE file in the reference's layout whatever the number of ranks -
//   int32 {total nvars, total nwcon, ncon}, mu, s, t, z, zs, zt (ncon each, written by rank 0), then x, zl, zu
//   as global arrays (each rank writes its block at its offset, as MPI_File_write_at_all does there), then zw
//   and sw likewise when there are sparse constraints.
// A file written by N ranks is byte-for-byte the file of the concatenated problem and can be read by any
// number of ranks (shared file system, like MPI-IO).
int InteriorPoint::gatherCounts(int64_t mine, std::vector<int64_t> *all) {
  // all[r] = the count of rank r, through the one collective the context has: a dot product of the
  // 1-vector [mine] with unit vectors that are 1 only on their own rank
  all->assign(ctx->size, mine);
  if (ctx->size == 1) return PO_OK;
  if (ctx->size > kMaxPanel) {
    set_error("solution files support at most %d ranks", kMaxPanel);
    return PO_ERR_ARG;
  }
  Vec *v = vec_new(ctx, 1), *one = vec_new(ctx, 1), *zero = vec_new(ctx, 1);
  if (!v || !one || !zero) return PO_ERR_HIP;
  int rc = k_fill(ctx, v->d, 1, (double)mine);
  if (rc == PO_OK) rc = k_fill(ctx, one->d, 1, 1.0);
  if (rc == PO_OK) rc = k_fill(ctx, zero->d, 1, 0.0);
  std::vector<const double *> V(ctx->size, zero->d);
  V[ctx->rank] = one->d;
  std::vector<double> out(ctx->size, 0.0);
  if (rc == PO_OK) rc = k_mdot(ctx, v->d, V.data(), ctx->size, 1, out.data());
  for (int r = 0; r < ctx->size; r++) (*all)[r] = (int64_t)(out[r] + 0.5);
  vec_decref(v);
  vec_decref(one);
  vec_decref(zero);
  return rc;
}

int InteriorPoint::solutionFileOffsets(int64_t *nvars_total, int64_t *var_off, int64_t *nw_total, int64_t *w_off) {
  std::vector<int64_t> nv, nwv;
  PO_TRY(gatherCounts(n, &nv));
  PO_TRY(gatherCounts(nw, &nwv));
  *nvars_total = *nw_total = *var_off = *w_off = 0;
  for (int r = 0; r < ctx->size; r++) {
    if (r < ctx->rank) {
      *var_off += nv[r];
      *w_off += nwv[r];
    }
    *nvars_total += nv[r];
    *nw_total += nwv[r];
  }
  return PO_OK;
}

static bool file_block(FILE *fp, bool writing, int64_t byte_offset, double *host, size_t count) {
  if (fseeko(fp, (off_t)byte_offset, SEEK_SET) != 0) return false;
  if (count == 0) return true;
  return (writing ? fwrite(host, sizeof(double), count, fp) : fread(host, sizeof(double), count, fp)) == count;
}

int InteriorPoint::writeSolutionFile(const char *filename) {
  int64_t N = 0, off = 0, Wt = 0, woff = 0;
  PO_TRY(solutionFileOffsets(&N, &off, &Wt, &woff));  // collective
  int ok = 1;
  if (ctx->rank == 0) {
    FILE *fp = fopen(filename, "wb");
    if (fp) {
      int sizes[3] = {(int)N, (int)Wt, c};
      ok = fwrite(sizes, sizeof(int), 3, fp) == 3 && fwrite(&barrier_param, sizeof(double), 1, fp) == 1;
      const std::vector<double> *blocks[5] = {&vars.s, &vars.t, &vars.z, &vars.zs, &vars.zt};
      for (auto *b : blocks) ok = ok && fwrite(b->data(), sizeof(double), c, fp) == (size_t)c;
      fclose(fp);
    } else {
      ok = 0;
    }
  }
  // every rank learns whether the header is there (and waits for it)
  std::vector<int64_t> flags;
  PO_TRY(gatherCounts(ok, &flags));
  if (flags[0] == 0) {
    set_error("cannot open checkpoint file %s", filename);
    return PO_ERR_ARG;
  }
  FILE *fp = fopen(filename, "r+b");
  if (!fp) {
    ok = 0;
  } else {
    const int64_t base = 3 * (int64_t)sizeof(int) + (5 * (int64_t)c + 1) * (int64_t)sizeof(double);
    std::vector<double> host((size_t)std::max<int64_t>(std::max(n, nw), 1));
    Vec *vs[3] = {x, zl, zu};
    for (int b = 0; b < 3 && ok; b++) {
      ok = hipMemcpyAsync(host.data(), vs[b]->d, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, ctx->stream) ==
               hipSuccess &&
           hipStreamSynchronize(ctx->stream) == hipSuccess &&
           file_block(fp, true, base + 8 * (b * N + off), host.data(), (size_t)n);
    }
    if (Wt > 0) {  // zw then sw, as the reference (:951-968)
      for (int b = 0; b < 2 && ok; b++) {
        ok = (nw == 0 || (hipMemcpyAsync(host.data(), wvar[b]->d, sizeof(double) * (size_t)nw,
                                         hipMemcpyDeviceToHost, ctx->stream) == hipSuccess &&
                          hipStreamSynchronize(ctx->stream) == hipSuccess)) &&
             file_block(fp, true, base + 8 * (3 * N + b * Wt + woff), host.data(), (size_t)nw);
      }
    }
    fclose(fp);
  }
  PO_TRY(gatherCounts(ok, &flags));  // completion barrier + agreement on the outcome
  for (int r = 0; r < ctx->size; r++) {
    if (flags[r] == 0) {
      set_error("writing the checkpoint file %s failed on rank %d", filename, r);
      return PO_ERR_ARG;
    }
  }
  return PO_OK;
}

// readSolutionFile (:983-1104): restart state written by writeSolutionFile (same layout, any rank count)
int InteriorPoint::readSolutionFile(const char *filename) {
  acz_valid = false;
  cwx_valid = false;
  int64_t N = 0, off = 0, Wt = 0, woff = 0;
  PO_TRY(solutionFileOffsets(&N, &off, &Wt, &woff));  // collective
  FILE *fp = fopen(filename, "rb");
  int ok = fp != nullptr;
  int sizes[3] = {0, 0, 0};
  bool size_ok = true;
  if (ok) {
    ok = fread(sizes, sizeof(int), 3, fp) == 3;
    size_ok = ok && sizes[0] == (int)N && sizes[1] == (int)Wt && sizes[2] == c;
    ok = ok && size_ok && fread(&barrier_param, sizeof(double), 1, fp) == 1;
    std::vector<double> *blocks[5] = {&vars.s, &vars.t, &vars.z, &vars.zs, &vars.zt};
    for (auto *b : blocks) ok = ok && fread(b->data(), sizeof(double), c, fp) == (size_t)c;
    const int64_t base = 3 * (int64_t)sizeof(int) + (5 * (int64_t)c + 1) * (int64_t)sizeof(double);
    std::vector<double> host((size_t)std::max<int64_t>(std::max(n, nw), 1));
    Vec *vs[3] = {x, zl, zu};
    for (int b = 0; b < 3 && ok; b++) {
      ok = file_block(fp, false, base + 8 * (b * N + off), host.data(), (size_t)n) &&
           hipMemcpyAsync(vs[b]->d, host.data(), sizeof(double) * (size_t)n, hipMemcpyHostToDevice, ctx->stream) ==
               hipSuccess &&
           hipStreamSynchronize(ctx->stream) == hipSuccess;
    }
    if (Wt > 0) {
      for (int b = 0; b < 2 && ok; b++) {
        ok = file_block(fp, false, base + 8 * (3 * N + b * Wt + woff), host.data(), (size_t)nw) &&
             (nw == 0 || (hipMemcpyAsync(wvar[b]->d, host.data(), sizeof(double) * (size_t)nw,
                                         hipMemcpyHostToDevice, ctx->stream) == hipSuccess &&
                          hipStreamSynchronize(ctx->stream) == hipSuccess));
      }
    }
    fclose(fp);
  }
  std::vector<int64_t> flags;
  PO_TRY(gatherCounts(ok, &flags));  // every rank takes the same decision
  for (int r = 0; r < ctx->size; r++) {
    if (flags[r] == 0) {
      if (!fp) {
        set_error("cannot open solution file %s", filename);
      } else if (!size_ok) {
        set_error("ParOpt: Problem size incompatible with solution file");
      } else {
        set_error("short read or upload failure on solution file %s (rank %d)", filename, r);
      }
      return PO_ERR_ARG;
    }
  }
  Vec *written[5] = {x, zl, zu, has_w ? wvar[0] : nullptr, has_w ? wvar[1] : nullptr};
  for (Vec *v : written) PO_TRY(refreshLiveMirror(v));
  return PO_OK;
}

}  // namespace po
