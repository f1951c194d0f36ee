// Host-side mirror of ParOptInteriorPoint (reference src/ParOptInteriorPoint.h:128-217) for the
// quasi-Newton branch: dense constraints, plus the sparse "weighting" constraints with nwblock = 1
// (ip_w.cpp).  Same public method names, option names/defaults and return codes; every n- or
// w-sized operation is a HIP kernel launch (core.hpp, wcon.hpp).
#pragma once
#include <stdlib.h>
#include <map>
#include <string>
#include <vector>

#include "problem.hpp"
#include "qn.hpp"

namespace po {

// Typed option registry with the names, defaults and ranges of
// ParOptInteriorPoint::addDefaultOptions (src/ParOptInteriorPoint.cpp:536-727).
class Options {
 public:
  enum Type { STR = 1, BOOL = 2, INT = 3, FLOAT = 4, ENUM = 5 };
  struct Entry {
    Type type;
    std::string s;
    int i = 0, ilo = 0, ihi = 0;
    double f = 0, flo = 0, fhi = 0;
    std::vector<std::string> choices;
  };
  Options();
  // ParOptTrustRegion::addDefaultOptions (src/ParOptTrustRegion.cpp:739-847) into the same registry
  void addTrustRegionDefaults();
  // ParOptMMA::addDefaultOptions (src/ParOptMMA.cpp:234-289)
  void addMMADefaults();
  int set(const char *name, const char *value);
  int set(const char *name, int value);
  int set(const char *name, double value);
  const char *str(const char *name) const;
  int integer(const char *name) const;
  double real(const char *name) const;
  bool has(const char *name) const { return e.count(name) > 0; }
  // calls fn for every entry that `skip` (may be null) does not hold: po_options_visit_defaults
  int visit(const Options *skip, po_option_visitor fn, void *user) const;
  // takes over every entry of `src` that `base` does not hold (the trust-region / MMA options of a driver whose
  // interior-point solver was created separately: one registry serves both, as in the reference)
  void adoptExtras(const Options &src, const Options &base);

 private:
  std::map<std::string, Entry> e;
};

struct Dense {  // the c-sized blocks of ParOptVars (src/ParOptInteriorPoint.h:373-389)
  std::vector<double> z, s, t, zs, zt;
  void resize(int c) {
    z.assign(c, 0.0);
    s.assign(c, 0.0);
    t.assign(c, 0.0);
    zs.assign(c, 0.0);
    zt.assign(c, 0.0);
  }
};

class InteriorPoint {
 public:
  InteriorPoint(Problem *prob);
  ~InteriorPoint();
  int allocate();  // device storage (fails with PO_ERR_HIP when HBM is exhausted)
  // SURVEY 8a' bookkeeping (po_ip_get_debug_ints)
  const std::vector<int> &gPivots() const { return gpiv; }
  Vec *lowerBounds() { return lb; }
  Vec *upperBounds() { return ub; }
  int checkFlag() const { return check_flag; }
  int clampCounts(double out[8]);
  // checkKKTStep (:6212-6360), step_verification_frequency: appends the block maxima of the linearised KKT
  // residual at the current (unscaled) step to the iteration history
  int checkKKTStep(int iteration, double mu);

  Options options;
  int optimize(const char *checkpoint);
  void getOptimizedPoint(Vec **x, const double **z, Vec **zl, Vec **zu);
  void getOptimizedSlacks(const double **s, const double **t, const double **zs, const double **zt);
  void getOptimizedSparse(Vec *v[5]);  // zw, sw, tw, zsw, ztw (null when nwcon = 0)
  void getIterationCounters(int *niter_, int *neval_, int *ngeval_);
  double getBarrierParameter() const { return barrier_param; }
  int getComplementarity(double *comp);
  void setPenaltyGamma(double gamma);
  void setPenaltyGammaArray(const double *gamma);  // per-constraint values (:1160-1172)
  int resetDesignAndBounds();
  void resetQuasiNewtonHessian();
  // drivers that own the quasi-Newton object / swap the problem (trust region): :1193-1234, :745-764
  int setQuasiNewton(CompactQuasiNewton *qn_);
  int resetProblemInstance(Problem *p);
  int writeSolutionFile(const char *filename);
  int readSolutionFile(const char *filename);
  int debugKKTStep(double mu);
  // State-injected known-answer tests (po_ip_debug_set_state / po_ip_debug_kkt): the caller has written x, zl, zu
  // (and the sparse blocks) through the borrowed handles; this takes the dense blocks and mu, evaluates the problem
  // at x (objective, constraints, gradient, Jacobian) and drops every quantity carried between passes.
  int debugSetState(const double *z, const double *s, const double *t, const double *zs, const double *zt, double mu);
  // mode 0: residual + Schur complements + ONE bordered solve, the stored-step kernels (computeKKTStep :2700-2737);
  // mode 1: the sequence of a plain quasi-Newton iteration of optimize() -- dinv_d1, the fused Gram pass (unformed
  //         L-SR1 columns), first solve pass with the refinement's products, refinement pass -- with the
  //         bound-multiplier steps stored.  Afterwards the matrices AS ASSEMBLED are kept in Gmat0 / Ce0.
  int debugKKT(double mu, int mode, double tau);
  std::vector<double> Gmat0, Ce0;   // G and Ce before their LU factorizations (filled inside debugKKT only)
  bool debug_keep_schur = false;
  double debug_norms[4] = {0, 0, 0, 0};  // max_prime, max_dual, max_infeas, res_norm of the last debugKKT
  const std::vector<double> &gramMatrix() const { return W; }
  const std::vector<int> &cPivots() const { return cpiv; }
  const double *stepMins() const { return step_mins; }
  Vec *dinvVec() { return Dinv; }
  Vec *rxVec() { return rx; }
  void flushHistory();
  // checkGradients(dh) (:6196-6199): the problem's finite-difference check at the current point; the report text
  // (what the reference prints) is returned
  int checkGradients(double dh, std::string *report);
  // checkMeritFuncGradient(xpt, dh) (:3280-3432): forward difference of the merit function along the current step
  // (xpt == nullptr) or, from xpt, along -g/|g| with the reference's fixed slack steps; out = {fd, actual}
  int checkMeritFuncGradient(Vec *xpt, double dh, double out[2]);

  Problem *prob;
  Ctx *ctx;
  int64_t n;
  int c;
  CompactQuasiNewton *qn;
  po_qn_s qn_handle;

  // state
  Vec *x, *zl, *zu, *lb, *ub, *g;
  std::vector<Vec *> Ac;
  Dense vars, res, step, refine;
  double fobj, barrier_param, rho_penalty_search;
  std::vector<double> cvals;
  int niter, neval, ngeval, nhvec;

  // observer + history
  bool analytic_panel_dots;  // debugging switch: false re-measures P^T px with an mdot pass
  po_ip_iteration_fn iter_cb;
  void *iter_cb_user;
  std::string history;
  std::vector<std::string> phase_names;
  std::vector<double> phase_seconds;
  std::string phase_names_joined;
  // Event timing of the problem's callbacks (the "user_eval" phase).  Off by default (round 6): every event record
  // is a packet between two kernels and costs the stream dispatch latency -- 850 / 853 against 863 / 866 inner it/s
  // at config 5 in one call (5-6 callbacks per 1.2 ms inner iteration), nothing measurable at n >= 10 M.
  // po_ip_set_callback_timing switches it on (bench.py does, for the figure it reports).
  bool user_timing = getenv("PAROPT_AMD_USER_TIMING") != nullptr;

  // step storage (exposed for the single-step known-answer tests)
  Vec *px, *pzl, *pzu;

  // sparse constraint blocks (nwblock = 1): zw, sw, tw, zsw, ztw of the variables / residual / step
  int64_t nw;        // local number of sparse constraints
  bool has_w;        // some rank has sparse constraints (uniform across ranks)
  double nw_global;  // total count (for the complementarity average)
  Vec *wvar[5], *wresv[5], *wstepv[5];

 private:
  // work vectors
  Vec *Dinv, *rx, *tvec, *xt, *y_qn, *s_qn;
  Vec *vA;  // A^T pz of the current (unscaled) step, accumulated by the solves
  std::vector<double> gamma_s, gamma_t;
  int use_lower, use_upper;
  bool qn_created;
  bool qn_owned;

  // small dense systems
  std::vector<double> W;            // (c+k)^2 weighted Gram, column-major
  int wk;                           // k used when W was assembled
  std::vector<double> Gf, Cef;      // LU factors
  std::vector<int> gpiv, cpiv;
  // residual bookkeeping of the last computeResidual call
  double comp_prod, comp_count, max_rx, max_rzl, max_rzu;
  double l1_rx = 0, l1_rzl = 0, l1_rzu = 0, l2_rx = 0, l2_rzl = 0, l2_rzu = 0;
  // lazily applied step scalings (scaleKKTStep :3253-3268)
  double sx, sz;
  double step_mins[2];
  // P^T px of the current (unscaled) step, P = [Ac | Z]: maintained analytically from the solves
  // (px = t + Dinv*(P alpha)  =>  P^T px = P^T t + W alpha), so that neither iterative refinement
  // nor the merit derivative needs another pass over the panel
  std::vector<double> ptpx;
  bool ptpx_valid;
  bool residual_fused;  // the last first-pass solve already wrote the refinement rhs t'
  bool residual_cached; // rx / norms of the CURRENT state were already evaluated (step update)
  bool corrector_active;  // Mehrotra predictor-corrector: s_qn / y_qn hold the corrector products
  // ... unless the corrector solve forms them itself (round 6: k_corr_d1_dots + k_solve2c, two launches and two host
  // round trips instead of five and three); corr_out: what k_solve2c reduced
  bool corrector_fused = false;
  // Dinv / t of the next first solve left behind by the residual pass of the new point (no quasi-Newton update in
  // between: round 6); valid for the diagonal and the barrier parameter recorded with them
  bool spec_dt_valid = false, spec_dt_want = false;
  double spec_dt_diag = 0.0, spec_dt_bmu = 0.0, spec_dt_mu = 0.0;
  double corr_out[12] = {0};
  int norm_type;          // 0 infinity, 1 l1, 2 l2 (ParOptNormType)
  std::vector<double> tdots;  // P^T t' produced by the fused first solve pass
  bool tdots_valid;
  bool fused_dots;        // use k_solve2_dots (switch PAROPT_AMD_NO_FUSED_DOTS=1 to compare)

  // ---- second-order information (ip_gmres.cpp) ----
  Vec *hdiag;                  // use_diag_hessian: diagonal of the Lagrangian Hessian (zero until evaluated)
  std::vector<Vec *> gmresW;   // Krylov basis of computeKKTGMRESStep
  bool vA_valid;               // vA = A^T pz of the current step was maintained by the solves
  bool inexact_newton_step;    // the current step came from computeKKTGMRESStep
  // merit pieces of the current (unscaled) step, produced together with the complementarity check of
  // scaleKKTStep: {pos log, neg log, ppos, pneg, g.px, px.px, max|px|}
  double merit_cache[7];
  double px_amax_w = 0.0;       // max|px| taken inside evalMeritInitDeriv's batch (sparse-constraint path)
  bool px_amax_valid = false;
  std::vector<double> Wt_buf, W2_buf;  // landing areas of Gram launches deferred to a batch flush (setUpKKTSystem)
  bool merit_cache_valid;
  // the same pieces taken by the refinement pass itself (k_solve2r with g): {S10, S01, S11, ppos, pneg, g.px, px.px,
  // max_x, max_z, max|px|}; the log-barrier sums of the iterate are those of the accepted trial point of the last
  // line search (same kernel arithmetic, same element order as the separate pass took them)
  double fused_merit[10] = {0};
  bool fused_merit_valid = false, fuse_merit = true;
  // "lean step" (round 3): in the plain quasi-Newton iteration the refinement pass stores px only; the bound-
  // multiplier steps are re-formed from it inside the multiplier update (kkt_res_update_kernel) -- two output streams
  // of the refinement pass less.  lean_step_allowed is set by optimize() around the step computation of an iteration
  // whose every later consumer of (pzl, pzu) is that update; pz_stored says whether the vectors hold the current step.
  // tvec / Dinv hold exactly what dinv_d1_kernel formed for (t0_diag, t0_mu) from the bound data and rx: a pass that
  // loads those anyway may re-form them in registers instead of reading them (k_solve2r with t1 == nullptr)
  bool t_is_plain_dinv_d1 = false, first_t_recomputable = false, recompute_dt = true;
  double t0_diag = 0.0;
  bool lean_step = true, lean_step_allowed = false, pz_stored = true;
  double step_beta_mu = 0.0;
  double trial_logs[2] = {0, 0}, iterate_logs[2] = {0, 0};
  bool trial_logs_valid = false, iterate_logs_valid = false;
  int gatherCounts(int64_t mine, std::vector<int64_t> *all);
  int solutionFileOffsets(int64_t *nvars_total, int64_t *var_off, int64_t *nw_total, int64_t *w_off);
  int ensureHdiag();
  int solveKKTAlpha(const double *bx, double alpha, const Dense &b, double mu, bool use_qn, bool full,
                    double tau, Dense &out);
  int evalObjBarrierDeriv(const Dense &p, double *pmerit);
  // sparse-constraint variants of the two above (ip_gmres.cpp); wscalev = alpha-scaled copy of the w residual
  int solveKKTAlphaW(const double *bx, double alpha, const Dense &b, double mu, bool use_qn, bool full,
                     double tau, Dense &out);
  Vec *wscalev[5];
  double w_merit_last[10];  // k_w_merit sums of the last evalObjBarrierDeriv (sparse part)
  int computeKKTGMRESStep(double rtol, double atol, bool use_qn, double tau, int *gmres_iters);

  // ---- sparse-constraint path (ip_w.cpp) ----
  Vec *gsw, *gtw, *Cw, *wd2, *wyw, *wtmp, *wtmp2;
  // cw(x) of the CURRENT iterate (round 4): the residual (twice per iteration: new barrier parameter), the
  // refinement's residual and the merit derivative all need the sparse constraint values at the same point; the
  // reference calls evalSparseCon each time (:1358, 3740), here the callback runs once per point and the accepted
  // trial point of the line search hands its values over.  Dropped whenever x or the problem instance changes.
  Vec *cwx = nullptr;
  bool cwx_valid = false, trial_cw_valid = false;
  int sparseConAtIterate(const double **cw);
  Vec *d1v;                 // n-sized: raw d1, then v = d1 + P alpha
  std::vector<Vec *> Uw;    // U_j = Aw (Dinv o P_j)
  bool panel_plain = false;  // Uw is the unscaled panel image (scalar block form)
  bool panel_valid = false;  // Uw matches the current setUpKKTSystem (consumed by solveKKTW)
  double w_sums[7], w_maxs[5];  // reductions of the last w residual (k_w_res layout)
  double res_out[13] = {0}, wres_out[12] = {0};  // landing area of the residual reductions (see after_reduce)
  // Monotone barrier strategy with the infinity norm (round 4): the residual pass also takes max|rzl|, max|rzu| for the
  // barrier parameter the strategy would switch to (a function of the current one alone), so that the switch needs
  // neither the mu-only pass over the bound data nor its host synchronisation (PAROPT_AMD_NO_SPEC_MU=1 restores it)
  bool spec_mu_on = true, spec_enabled = false, spec_valid = false;
  double spec_mu = 0.0, spec_max[2] = {0.0, 0.0};
  double nextMonotoneMu() const;
  WVars wv() const;
  WVars wr() const;
  WVars wp() const;
  int allocateW();
  int applyK0(const double *bx, const double *bw, Vec *yx, Vec *yw);
  // d2out: also wd2 = d2 of the block solve that follows (one launch less); wd2_ready tells solveKKTW
  int computeResidualW(double mu, bool norms = true, bool with_d2 = false);
  bool wd2_ready = false;
  int sparseGramCorrection(const std::vector<const double *> &P, int m, Vec *work = nullptr, bool may_defer = false,
                           bool panel_done = false);
  int panelImageVectors(int m, std::vector<double *> &U);  // Uw[0..m) as raw pointers (allocated on demand)
  int solveKKTW(const Dense &b, double mu, bool use_qn, bool refine_pass, double tau, Dense &out,
                bool fuse_residual = false);
  int computeKKTStepWithRefinementW(double mu, bool use_qn, double tau);
  int initLeastSquaresMultipliersW();
  int wCompStep(double ax, double az, double *prod);

  Bounds bounds() const;
  std::vector<const double *> panel(bool use_qn, int *k) const;

  int createQuasiNewton();
  int initAndCheckDesignAndBounds();
  int initLeastSquaresMultipliers();
  int initAffineStepMultipliers();
  void denseResidual(double mu, Dense &r) const;
  struct MultUpdate {  // a bound-multiplier step that rides in the residual pass (computeStepAndUpdate)
    double a = 0.0, az = 0.0, eps = 0.0;
    bool acz_follow = false;
  };
  int computeResidual(double mu, bool vectors, Vec *yqn_complete = nullptr, const MultUpdate *upd = nullptr);
  void resNorms(const Dense &r, double *max_prime, double *max_dual, double *max_infeas,
                double *res_norm) const;
  double compFromSums(double prod, double count, const Dense &v, double wprod = 0.0) const;
  // diag_only: keep b0 of the quasi-Newton approximation in Dinv but leave its low-rank part out
  // (the "diagonal quasi-Newton step" of :4923-4980)
  int setUpKKTSystem(bool use_qn, bool diag_only = false, const double *rhs_mu = nullptr);
  int solveKKT(const Dense &b, double mu, bool use_qn, bool refine_pass, double tau, Dense &out,
               bool fuse_residual = false);
  int computeKKTStepWithRefinement(double mu, bool use_qn, double tau);
  int mehrotraStep(bool use_qn, double comp, bool corrector, double min_frac, double abs_res_tol, double *tau_out);
  int scaleKKTStep(double tau, double comp, double *alpha_x, double *alpha_z, int *ceq,
                   bool inexact = false);
  int evalMeritInitDeriv(double max_x, double *merit, double *pmerit);
  double evalMeritFromSums(double fk, const double *ck, const double *sk, const double *tk,
                           double pos, double neg, const double *wsums = nullptr) const;
  int lineSearch(double alpha_min, double *alpha, double m0, double dm0, int *fail);
  int computeStepAndUpdate(double alpha, int eval_obj_con, int perform_qn_update, int *update_type);
  void phaseBegin();
  void phaseEnd(const char *name);
  double phase_t0;
  // HIP-event time of the user's problem callbacks (evalObjCon / evalObjConGradient): reported as the phase
  // "user_eval", a SUBSET of the init / line_search / step_update phases that contain the calls
  static const int kUserRing = 16;
  hipEvent_t user_ev[2 * kUserRing];
  int user_pending = 0;
  bool user_events_ready = false;
  double user_seconds = 0.0;
  void userBegin();
  void userEnd();
  void userHarvest();
  int check_flag = 0;  // OR of the bound-repair bits of every initAndCheckDesignAndBounds call (:4290-4344)
  bool ac_valid = false;
  // A^T z of a problem with linear dense constraints, kept by recurrence (computeResidual / computeStepAndUpdate)
  static const int kAczRefresh = 16;
  Vec *acz = nullptr;
  bool acz_valid = false, use_acz = true, use_ztpx_hint = true;
  int acz_age = 0;
  // P^T t of the first solve, produced by the Gram pass of setUpKKTSystem (see there)
  bool fuse_mult_update = true;
  bool fast_yqn_w = true;  // sparse constraints: y_qn from the residuals (PAROPT_AMD_NO_FAST_YQN_W=1 restores the passes)
  // sparse constraints, round 4 ("lean" solve passes, PAROPT_AMD_NO_W_LEAN=1 restores the stored forms): the fused
  // first pass stores px only (px_first_only), the refinement pass re-forms the first bound-multiplier steps from it,
  // takes the complementarity / merit sums of the final step (fused_merit, as on the dense path) and, in the plain
  // quasi-Newton iteration, stores px only again; the sparse blocks' share of the complementarity polynomial comes
  // out of the step kernel (w_comp_poly) and the sparse merit sums ride in the same batch (w_merit_cache, sx = 1)
  bool w_lean = true, px_first_only = false;
  double w_step_out[5] = {0, 0, 0, 0, 0}, w_comp_poly[3] = {0, 0, 0}, w_merit_cache[10] = {0};
  bool w_comp_valid = false, w_merit_cache_valid = false;
  bool s_qn_from_trial = false;  // s_qn holds s_qn_a * px, written by the last trial pass of the line search
  double s_qn_a = 0.0;
  bool recompute_first_step = true, step_deferred = false;  // see solveKKT: the first pass stores no step
  std::vector<double> alpha_first, coef_first;  // coefficients of that first pass (solve, refinement residual)
  double diag_first = 0.0;
  bool recompute_rhs = true;  // ... and the refinement right-hand side is recomputed as well
  // unformed L-SR1 columns are consumed unformed by the Gram pass and both solve passes (never written)
  bool virtual_z = true, virt_first = false, allow_virtual_z = false;
  bool fused_tdots = true, t0_valid = false;
  double t0_mu = 0.0;
  std::vector<double> t0dots;  // Ac holds the Jacobian of a problem with linear_constraints
};

}  // namespace po

struct po_ip_s {
  po::InteriorPoint *ip;
};
