// The sparse ("weighting") constraint path of the interior-point mirror: nwcon > 0 with nwblock = 1,
// i.e. a diagonal Cw = (Cdiag + Aw D^-1 Aw^T)^-1 (reference src/ParOptSparseMat.cpp:41-229 and the
// `nwcon > 0` branches of src/ParOptInteriorPoint.cpp cited per function).
//
// The design-space operator of K0 changes from diag(Dinv) to
//     D0x^-1 = Dinv - Dinv Aw^T Cw Aw Dinv                      (ParOptQuasiDefBlockMat::apply :122-190)
// and everything the dense path does with the weighted Gram  W = P^T Dinv P  carries over with
//     W = P^T Dinv P - U^T Cw U,   U_j = Aw (Dinv o P_j)        (one extra w-sized Gram)
// so the Schur complements G, Ce and the analytic P^T px bookkeeping are unchanged (ip.cpp).
#include <math.h>
#include <string.h>

#include <algorithm>

#include "ip.hpp"

namespace po {

WVars InteriorPoint::wv() const { return WVars{wvar[0]->d, wvar[1]->d, wvar[2]->d, wvar[3]->d, wvar[4]->d}; }
WVars InteriorPoint::wr() const {
  return WVars{wresv[0]->d, wresv[1]->d, wresv[2]->d, wresv[3]->d, wresv[4]->d};
}
WVars InteriorPoint::wp() const {
  return WVars{wstepv[0]->d, wstepv[1]->d, wstepv[2]->d, wstepv[3]->d, wstepv[4]->d};
}

void InteriorPoint::getOptimizedSparse(Vec *v[5]) {
  for (int i = 0; i < 5; i++) v[i] = has_w ? wvar[i] : nullptr;
}

int InteriorPoint::allocateW() {
  // path selection must be uniform across ranks (every reduction is collective)
  Vec *cnt = vec_new(ctx, 1);
  if (!cnt) return PO_ERR_HIP;
  PO_TRY(k_fill(ctx, cnt->d, 1, (double)nw));
  double total = 0.0;
  PO_TRY(k_reduce1(ctx, RED_ASUM, cnt->d, nullptr, 1, &total));
  vec_decref(cnt);
  nw_global = total;
  has_w = total > 0.0;
  if (!has_w) return PO_OK;
  Vec **all[] = {&gsw, &gtw, &Cw, &wd2, &wyw, &wtmp, &wtmp2, &cwx};
  for (Vec **v : all) {
    *v = vec_new(ctx, nw);
    if (!*v) return PO_ERR_HIP;
  }
  for (int i = 0; i < 5; i++) {
    wvar[i] = vec_new(ctx, nw);
    wresv[i] = vec_new(ctx, nw);
    wstepv[i] = vec_new(ctx, nw);
    if (!wvar[i] || !wresv[i] || !wstepv[i]) return PO_ERR_HIP;
    PO_TRY(k_fill(ctx, wvar[i]->d, nw, 1.0));  // constructor state (:439-446)
  }
  d1v = vec_new(ctx, n);
  if (!d1v) return PO_ERR_HIP;
  PO_TRY(k_w_gamma(ctx, gsw->d, gtw->d, options.real("penalty_gamma"), prob->nwinequality, nw));
  return PO_OK;
}

// (yx, yw) <- K0^-1 (bx, bw), ParOptQuasiDefBlockMat::apply (src/ParOptSparseMat.cpp:122-190);
// bw == nullptr stands for a zero block.  bx must not alias yx->d.
int InteriorPoint::applyK0(const double *bx, const double *bw, Vec *yx, Vec *yw) {
  return prob->sparseApplyK0(x, Dinv, Cw, bx, bw, yx, yw, wtmp);
}

// the w blocks of computeKKTRes (:1358-1398) and their norms
// (norms = false: the blocks only -- the caller knows that the sums of this mu and iterate are already in place)
int InteriorPoint::sparseConAtIterate(const double **cw) {
  if (!cwx_valid) {
    if (prob->evalSparseCon(x, cwx) != 0) return PO_ERR_USER;
    cwx_valid = true;
  }
  *cw = cwx->d;
  return PO_OK;
}

int InteriorPoint::computeResidualW(double mu, bool norms, bool with_d2) {
  const double *cw = nullptr;
  PO_TRY(sparseConAtIterate(&cw));
  wd2_ready = false;
  PO_TRY(k_w_res(ctx, wv(), wr(), gsw->d, gtw->d, mu, nw, norms ? wres_out : nullptr, cw, with_d2 ? wd2->d : nullptr));
  if (!norms) return PO_OK;
  after_reduce(ctx, [this] {
    for (int i = 0; i < 7; i++) w_sums[i] = wres_out[i];
    for (int i = 0; i < 5; i++) w_maxs[i] = wres_out[7 + i];
  });
  return PO_OK;
}

int InteriorPoint::wCompStep(double ax, double az, double *prod) {
  return k_w_comp_step(ctx, wv(), wp(), ax, az, nw, prod);
}

// W -= U^T S^-1 U with U_j = Aw (Dinv o P_j), S = C + Aw Dinv Aw^T (diagonal for the block form: Cw = S^-1)
int InteriorPoint::panelImageVectors(int m, std::vector<double *> &U) {
  while ((int)Uw.size() < m) {
    Vec *u = vec_new(ctx, nw);
    if (!u) return PO_ERR_HIP;
    Uw.push_back(u);
  }
  U.resize(m);
  for (int j = 0; j < m; j++) U[j] = Uw[j]->d;
  return PO_OK;
}
// panel_done: the Gram pass over P has already written U (Problem::sparseGramGroups)
int InteriorPoint::sparseGramCorrection(const std::vector<const double *> &P, int m, Vec *work, bool may_defer,
                                        bool panel_done) {
  if (m <= 0) return PO_OK;
  std::vector<double *> U;
  PO_TRY(panelImageVectors(m, U));
  std::vector<const double *> Uc(U.begin(), U.end());
  if (!panel_done) PO_TRY(prob->sparseJacobianPanel(x, Dinv, P.data(), m, U.data(), work ? work : tvec));
  // block form: U^T Cw U.  CSR form: U <- L^-1 U with S = L L^T, then U^T U
  const double *weights = Cw->d;
  PO_TRY(prob->sparseHalfSolve(U.data(), m, Cw, &weights));
  panel_plain = (weights == Cw->d);  // scalar block form: Uw still holds U = Aw (Dinv o P) itself
  // (may_defer: inside setUpKKTSystem's batch the product arrives at the flush: a member holds it, the subtraction
  // follows it there)
  W2_buf.assign((size_t)m * m, 0.0);
  PO_TRY(k_wgram(ctx, weights, Uc.data(), m, nw, W2_buf.data(), nullptr, nullptr, 0, 0.0, 0, may_defer));
  after_reduce(ctx, [this] {
    for (size_t i = 0; i < W2_buf.size() && i < W.size(); i++) W[i] -= W2_buf[i];
  });
  panel_valid = true;  // Uw holds the (half-solved) panel of the CURRENT Dinv, factor and panel columns
  return PO_OK;
}

// computeKKTStep (:2700-2737) with both solveKKTDiagSystem overloads folded together, sparse blocks
// included.  first pass: rhs = (rx, wres, b); refine pass: d1v already holds the raw d1' and wres r'.
int InteriorPoint::solveKKTW(const Dense &b, double mu, bool use_qn, bool refine_pass, double tau,
                             Dense &out, bool fuse_residual) {
  const double beta_mu = options.real("rel_bound_barrier") * mu;
  int k = 0;
  std::vector<const double *> P = panel(use_qn, &k);
  if (k != wk) {
    set_error("internal: panel width changed between setUpKKTSystem and solve (%d vs %d)", k, wk);
    return PO_ERR_ARG;
  }
  const int m = c + k;
  const double *cl = (corrector_active && !refine_pass) ? s_qn->d : nullptr;
  const double *cu = (corrector_active && !refine_pass) ? y_qn->d : nullptr;
  // t = [K0^-1 (d1, d2)]_x, wyw and P^T t were produced by setUpKKTSystem when the right-hand side was known then
  const bool have_t0 = !refine_pass && t0_valid && t0_mu == mu && !cl && (int)t0dots.size() == m && m > 0;
  std::vector<double> dots(m > 0 ? m : 1, 0.0);
  if (have_t0) {
    for (int i = 0; i < m; i++) dots[i] = t0dots[i];
  } else {
    if (!refine_pass) PO_TRY(k_d1(ctx, bounds(), rx->d, nullptr, beta_mu, n, d1v->d, cl, cu));
    if (!wd2_ready) PO_TRY(k_w_d2(ctx, wv(), wr(), nw, wd2->d));  // (else: formed by the pass that wrote the blocks)
    wd2_ready = false;
    PO_TRY(applyK0(d1v->d, wd2->d, tvec, wyw));
    if (refine_pass && tdots_valid && (int)tdots.size() == m && m > 0 && panel_valid && panel_plain &&
        (int)Uw.size() >= m) {
      // tvec = Dinv o (d1' + Aw^T yw): P^T tvec = P^T (Dinv o d1') + U^T yw with U = Aw (Dinv o P) -- the first
      // term came out of the fused first pass, the second is a w-sized product with the panel image
      std::vector<const double *> Uc(m);
      for (int j = 0; j < m; j++) Uc[j] = Uw[j]->d;
      PO_TRY(k_mdot(ctx, wyw->d, Uc.data(), m, nw, dots.data()));
      for (int i = 0; i < m; i++) dots[i] += tdots[i];
    } else if (m > 0) {
      PO_TRY(k_mdot(ctx, tvec->d, P.data(), m, n, dots.data()));
    }
  }
  if (!refine_pass) t0_valid = false;  // tvec is overwritten below
  std::vector<double> yz(c > 0 ? c : 1, 0.0), yz2(c > 0 ? c : 1, 0.0), zeta(k > 0 ? k : 1, 0.0);
  for (int i = 0; i < c; i++) {
    yz[i] = (b.z[i] + (b.zs[i] + vars.s[i] * b.s[i]) / vars.zs[i] -
             (b.zt[i] + vars.t[i] * b.t[i]) / vars.zt[i] - dots[i]);
  }
  if (c > 0) lu_solve(c, Gf.data(), c, gpiv.data(), yz.data());
  if (k > 0) {
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
  w_comp_valid = false;
  w_merit_cache_valid = false;
  pz_stored = true;
  bool first_px_only = refine_pass && px_first_only;  // px holds the first step, pzl / pzu were not stored
  px_first_only = false;
  if (first_px_only && !(m > 0 && (int)Uw.size() >= m && panel_valid)) {
    // (not reached with the flags as they are set: the fused first pass implies a valid panel) the stored-step forms
    // below need the first bound-multiplier steps as vectors
    PO_TRY(k_form_pz(ctx, bounds(), px->d, beta_mu, n, pzl->d, pzu->d));
    first_px_only = false;
  }
  tdots_valid = false;        // (consumed above by a refinement pass; set again below by a fused first pass)
  if (!refine_pass) residual_fused = false;
  // (dx, dzw) = K0^-1 (d1 + P alpha, d2) = K0^-1 (d1, d2) + K0^-1 (P alpha, 0), and the second term comes from
  // the panel the Gram correction already holds: dzw += -S^-1 (U alpha), dx += Dinv (P alpha + Aw^T of that) -
  // no second quasi-definite apply, and P alpha rides in the same pass that forms the bound multipliers
  double mins_x[2], mins_w[2];
  bool w_done = false;  // the refinement branch below already took the sparse step with its other sums
  // the fraction-to-boundary minima of the design and of the sparse blocks: one collective + sync (opened right
  // before the first of the two launches: no user code runs between them)
  BatchScope minbatch(ctx, false);
  const bool seq_lin = options.integer("sequential_linear_method");
  const int kq = (qn && !seq_lin) ? qn->size() : 0;
  // Fused refinement residual (as on the dense path, ip.cpp solveKKT): the coefficients of addKKTResStep are known
  // before the axpy pass starts, and so is the sparse multiplier step pzw (it depends on wyw only), so ONE pass
  // over P writes the step, the RAW right-hand side d1' of the refinement solve (its design rows, :1451-1483, with
  // the extra column Aw^T pzw) and the panel products of Dinv o d1'.
  const bool fuse = fuse_residual && !refine_pass && fused_dots && analytic_panel_dots && kq == k && m > 0 &&
                    (int)Uw.size() >= m && panel_valid && panel_plain && !cl &&
                    !(options.integer("use_diag_hessian") && hdiag) && m + 2 <= kMaxPanel;
  if (fuse) {
    std::vector<const double *> Uc(m);
    for (int j = 0; j < m; j++) Uc[j] = Uw[j]->d;
    PO_TRY(prob->sparseCorrection(Uc.data(), m, alpha.data(), Cw, wtmp2, wyw));  // ... and wyw += wtmp2
    // the two extra columns Aw^T wtmp2 (coefficient 1 in the step's row sum) and Aw^T pzw (1 in the residual's): n-sized
    // vectors from Problem::setSparseJacobianTranspose, or -- structured problems -- described and formed by the pass
    GroupCols2 gcs;
    static const bool s2d_grouped = getenv("PAROPT_AMD_GROUP_COLS_S2D") ? atoi(getenv("PAROPT_AMD_GROUP_COLS_S2D")) != 0 : true;
    const bool grouped = s2d_grouped && prob->sparseTransposeColumn(1.0, x, wtmp2, &gcs.g[0]);
    if (!grouped && prob->setSparseJacobianTranspose(1.0, x, wtmp2, d1v) != 0) return PO_ERR_USER;
    // sparse blocks of the step first: pzw = wstepv[0] feeds the residual column Aw^T pzw (its minima wait for
    // those of the design blocks unless user code runs in between)
    if (prob->reductionsBatchable()) minbatch.begin();
    PO_TRY(k_w_step(ctx, wv(), wr(), wyw->d, 0, tau, wp(), nw, mins_w));
    if (grouped) {
      if (!prob->sparseTransposeColumn(1.0, x, wstepv[0], &gcs.g[1])) {
        set_error("internal: the problem describes one sparse transpose column but not the other");
        return PO_ERR_ARG;
      }
      gcs.count = 2;
      gcs.ca[0] = 1.0;
      gcs.cb[1] = 1.0;
    } else if (prob->setSparseJacobianTranspose(1.0, x, wstepv[0], xt) != 0) {
      return PO_ERR_USER;
    }
    std::vector<const double *> P1(P);
    std::vector<double> a1(alpha.begin(), alpha.begin() + m), c2(m + 2, 0.0);
    if (!grouped) {
      P1.push_back(d1v->d);
      a1.push_back(1.0);
      P1.push_back(xt->d);
      a1.push_back(0.0);
    }
    const int np1 = (int)P1.size();
    for (int i = 0; i < c; i++) c2[i] = alpha[i];  // = step.z
    double diag = options.real("qn_sigma");
    if (qn && !seq_lin) {
      diag += qn->diag();
      if (k > 0) {
        std::vector<double> rz(ptpx.begin() + c, ptpx.begin() + c + k);
        qn->applyCompactInverse(rz.data());
        for (int j = 0; j < k; j++) c2[c + j] = rz[j];
      }
    }
    c2[m + 1] = 1.0;
    std::vector<double> so(m + 4, 0.0);
    // the raw right-hand side lands in y_qn (free while no corrector is active: y_qn is rebuilt from scratch by
    // computeStepAndUpdate) -- d1v is one of the columns being read -- and the two exchange buffers afterwards.
    // Of the step only px is stored (w_lean): the refinement's sparse rows need Aw px as a vector, the bound-
    // multiplier steps are re-formed from px by the refinement pass (two output streams less)
    const int sstep = w_lean ? 2 : 1;
    PO_TRY(k_solve2_dots(ctx, bounds(), tvec->d, Dinv->d, a1.data(), c2.data(), P1.data(), np1, beta_mu, tau,
                         rx->d, diag, n, px->d, pzl->d, pzu->d, nullptr, nullptr, 0, so.data(), y_qn->d, sstep, 0,
                         nullptr, 0, 0.0, 0.0, grouped ? &gcs : nullptr));
    px_first_only = sstep == 2;
    std::swap(d1v->d, y_qn->d);
    PO_TRY(minbatch.end());
    tdots.assign(so.begin(), so.begin() + m);
    tdots_valid = true;
    residual_fused = true;
    mins_x[0] = so[np1];  // out = {dots[np1], max_x, max_z}
    mins_x[1] = so[np1 + 1];
  } else if (m > 0 && (int)Uw.size() >= m && panel_valid) {
    std::vector<const double *> Uc(m);
    for (int j = 0; j < m; j++) Uc[j] = Uw[j]->d;
    PO_TRY(prob->sparseCorrection(Uc.data(), m, alpha.data(), Cw, wtmp2, wyw));  // ... and wyw += wtmp2
    // the extra column Aw^T wtmp2 (coefficient 1): stored, or (structured problems, refinement of a px-only first
    // step) described and formed by the pass itself
    GroupCol gcol;
    const bool grouped = first_px_only && prob->sparseTransposeColumn(1.0, x, wtmp2, &gcol);
    if (!grouped && prob->setSparseJacobianTranspose(1.0, x, wtmp2, d1v) != 0) return PO_ERR_USER;
    std::vector<const double *> P1(P);
    std::vector<double> a1(alpha.begin(), alpha.begin() + m);
    if (!grouped) {
      P1.push_back(d1v->d);
      a1.push_back(1.0);
    }
    minbatch.begin();
    if (first_px_only) {
      // Refinement on top of a first step of which only px was stored: solve2r_kernel in its stored right-hand side
      // form with a ZERO first coefficient set and t1 = px -- it re-forms (pzl, pzu) of the first step from px
      // (the very expressions the first pass evaluated: same bits), applies the refinement (t2 = tvec, a2) and takes
      // the complementarity / merit sums of the FINAL step (fused_merit); the final px goes to xt (free during the
      // solves) and the two buffers are exchanged.  Lean step: (pzl, pzu) are not stored either -- their only
      // consumer left is the multiplier update of computeStepAndUpdate, which re-forms them (kkt_res_update_kernel)
      const bool take_merit = fuse_merit && !cl;
      const bool lean = take_merit && lean_step && lean_step_allowed && iterate_logs_valid && fast_yqn_w &&
                        !prob->linear_constraints && options.integer("iterative_refinement_steps") == 1;
      std::vector<double> azero(m + 1, 0.0);
      PO_TRY(k_solve2r(ctx, bounds(), px->d, tvec->d, Dinv->d, azero.data(), a1.data(), P1.data(), (int)P1.size(), beta_mu,
                       tau, n, xt->d, lean ? nullptr : pzl->d, lean ? nullptr : pzu->d, nullptr, 0, mins_x, nullptr,
                       nullptr, 0.0, 0, nullptr, 0, 0.0, take_merit ? g->d : nullptr,
                       take_merit ? fused_merit : nullptr, 0.0, grouped ? &gcol : nullptr, 0.0, 1.0));
      std::swap(px->d, xt->d);
      if (lean) {
        pz_stored = false;
        step_beta_mu = beta_mu;
      }
      if (take_merit) {
        // the sparse blocks of the final step, their share of the complementarity polynomial and the sparse merit
        // sums (at sx = 1) in the same batch: scaleKKTStep and evalMeritInitDeriv then launch nothing
        PO_TRY(k_w_step(ctx, wv(), wr(), wyw->d, 1, tau, wp(), nw, w_step_out, 1));
        const double *cw = nullptr;
        PO_TRY(sparseConAtIterate(&cw));
        PO_TRY(prob->setSparseJacobian(1.0, x, px, wtmp2));
        PO_TRY(k_w_merit(ctx, wv(), wp(), 1.0, gsw->d, gtw->d, cw, wtmp2->d, nw, w_merit_cache));
        PO_TRY(minbatch.end());
        mins_x[0] = fused_merit[7];
        mins_x[1] = fused_merit[8];
        mins_w[0] = w_step_out[3];
        mins_w[1] = w_step_out[4];
        for (int i = 0; i < 3; i++) w_comp_poly[i] = w_step_out[i];
        fused_merit_valid = true;
        w_comp_valid = true;
        w_merit_cache_valid = true;
        w_done = true;
      }
    } else {
      PO_TRY(k_solve2(ctx, bounds(), tvec->d, Dinv->d, a1.data(), P1.data(), m + 1, beta_mu, refine_pass ? 1 : 0,
                      tau, n, px->d, pzl->d, pzu->d, mins_x, nullptr, rx->d, 0.0, nullptr, nullptr, 0, cl, cu));
    }
  } else {
    if (m > 0) PO_TRY(k_panel_axpy(ctx, d1v->d, 0.0, nullptr, 1.0, alpha.data(), P.data(), m, n));
    PO_TRY(applyK0(d1v->d, wd2->d, tvec, wyw));
    minbatch.begin();
    PO_TRY(k_solve2(ctx, bounds(), tvec->d, Dinv->d, nullptr, nullptr, 0, beta_mu, refine_pass ? 1 : 0,
                    tau, n, px->d, pzl->d, pzu->d, mins_x, nullptr, rx->d, 0.0, nullptr, nullptr, 0, cl,
                    cu));
  }
  if (!fuse && !w_done) {
    PO_TRY(k_w_step(ctx, wv(), wr(), wyw->d, refine_pass ? 1 : 0, tau, wp(), nw, mins_w));
    PO_TRY(minbatch.end());
  }
  step_mins[0] = std::min(mins_x[0], mins_w[0]);
  step_mins[1] = std::min(mins_x[1], mins_w[1]);
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

int InteriorPoint::computeKKTStepWithRefinementW(double mu, bool use_qn, double tau) {
  const int nref = options.integer("iterative_refinement_steps");
  const double beta_mu = options.real("rel_bound_barrier") * mu;
  const bool seq_lin = options.integer("sequential_linear_method");
  if (!(t0_valid && t0_mu == mu)) PO_TRY(computeResidualW(mu));  // (setUpKKTSystem did it for the fused first solve)
  denseResidual(mu, res);
  PO_TRY(solveKKTW(res, mu, use_qn, false, tau, step, nref > 0));
  for (int it = 0; it < nref; it++) {  // :4985-4991
    int kq = 0;
    std::vector<const double *> Pq = panel(qn && !seq_lin, &kq);
    const int mq = c + kq;
    std::vector<double> dots(mq > 0 ? mq : 1, 0.0);
    if (analytic_panel_dots && ptpx_valid && mq == c + wk) {
      for (int i = 0; i < mq; i++) dots[i] = ptpx[i];
    } else if (mq > 0) {
      PO_TRY(k_mdot(ctx, px->d, Pq.data(), mq, n, dots.data()));
    }
    double diag = options.real("qn_sigma");
    std::vector<double> coef(mq + 2, 0.0);
    for (int i = 0; i < c; i++) coef[i] = step.z[i];
    int mres = mq;
    if (options.integer("use_diag_hessian") && hdiag) {
      // -h o px replaces the whole quasi-Newton term, sigma included (:1464-1473)
      diag = 0.0;
      PO_TRY(k_mul(ctx, xt->d, 1.0, hdiag->d, px->d, n));
      Pq.push_back(xt->d);
      coef[mres++] = -1.0;
    } else if (qn && !seq_lin) {
      diag += qn->diag();
      if (kq > 0) {
        std::vector<double> rz(dots.begin() + c, dots.begin() + c + kq);
        qn->applyCompactInverse(rz.data());
        for (int j = 0; j < kq; j++) coef[c + j] = rz[j];
      }
    }
    // addKKTResStep (:1451-1583): design rows with the extra column Aw^T pzw, raw d1' into d1v (already there
    // when the first solve ran in its fused form)
    if (!(it == 0 && residual_fused)) {
      if (prob->setSparseJacobianTranspose(1.0, x, wstepv[0], tvec) != 0) return PO_ERR_USER;
      Pq.push_back(tvec->d);
      coef[mres++] = 1.0;
      PO_TRY(k_res_step(ctx, bounds(), rx->d, px->d, pzl->d, pzu->d, nullptr, coef.data(), Pq.data(),
                        mres, diag, beta_mu, n, d1v->d));
    }
    // sparse rows (:1492-1527); only the blocks are rebuilt here.  w_sums / w_maxs are deliberately LEFT at the norms
    // of the iterate's own barrier parameter (the affine solve of the Mehrotra strategies runs this with mu = 0, and
    // nothing reads the norms before the next computeResidual evaluates them again)
    PO_TRY(computeResidualW(mu, false));
    if (prob->addSparseJacobian(-1.0, x, px, wresv[0]) != 0) return PO_ERR_USER;
    PO_TRY(k_w_res_step(ctx, wv(), wp(), wr(), nw, wd2->d));  // ... and d2 of the refinement's block solve
    wd2_ready = true;
    Dense r2;
    r2.resize(c);
    denseResidual(mu, r2);
    for (int i = 0; i < c; i++) {  // dense rows :1529-1535
      r2.z[i] -= (dots[i] - step.s[i] + step.t[i]);
      r2.s[i] += (step.zs[i] - step.z[i]);
      r2.t[i] += (step.zt[i] + step.z[i]);
      r2.zs[i] -= (step.s[i] * vars.zs[i] + vars.s[i] * step.zs[i]);
      r2.zt[i] -= (step.t[i] * vars.zt[i] + vars.t[i] * step.zt[i]);
    }
    PO_TRY(solveKKTW(r2, mu, use_qn, true, tau, refine));
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

// initLeastSquaresMultipliers (:5366-5534) with the sparse blocks
int InteriorPoint::initLeastSquaresMultipliersW() {
  const double mu0 = options.real("init_barrier_param");
  for (int i = 0; i < 5; i++) PO_TRY(k_fill(ctx, wvar[i]->d, nw, mu0));
  const double small = 1e-4;
  PO_TRY(k_fill(ctx, Dinv->d, n, 1.0));
  PO_TRY(k_fill(ctx, Cw->d, nw, small));
  PO_TRY(prob->sparseFactor(x, Dinv, Cw));  // mat->factor (:5429)
  int k = 0;
  std::vector<const double *> A = panel(false, &k);
  W.assign((size_t)c * c, 0.0);
  std::vector<int> piv(c > 0 ? c : 1);
  if (c > 0) {
    PO_TRY(k_wgram(ctx, Dinv->d, A.data(), c, n, W.data()));
    PO_TRY(sparseGramCorrection(A, c, nullptr));
    for (int i = 0; i < c; i++) W[(size_t)i * (c + 1)] += small;
    lu_factor(c, W.data(), c, piv.data());
  }
  // rx = -(g - zl + zu)
  const double al[2] = {1.0, -1.0};
  const double *vv[2] = {zl->d, zu->d};
  PO_TRY(k_panel_axpy(ctx, d1v->d, -1.0, g->d, 0.0, al, vv, 2, n));
  PO_TRY(applyK0(d1v->d, nullptr, tvec, wyw));
  std::vector<double> z(c > 0 ? c : 1, 0.0);
  if (c > 0) {
    PO_TRY(k_mdot(ctx, tvec->d, A.data(), c, n, z.data()));
    for (int i = 0; i < c; i++) z[i] = -z[i];
    lu_solve(c, W.data(), c, piv.data(), z.data());
    PO_TRY(k_panel_axpy(ctx, d1v->d, 0.0, nullptr, 1.0, z.data(), A.data(), c, n));
  }
  PO_TRY(applyK0(d1v->d, nullptr, tvec, wyw));
  for (int i = 0; i < c; i++) {
    const double gam = 10.0 * std::max(gamma_s[i], gamma_t[i]);
    vars.z[i] = (z[i] < -gam || z[i] > gam) ? 0.0 : z[i];
  }
  PO_TRY(k_w_clip(ctx, wvar[0]->d, wyw->d, gsw->d, gtw->d, nw));
  panel_valid = false;  // Uw was built with Dinv = 1 and the constraint columns only
  return PO_OK;
}

}  // namespace po
