// Second-order branches of the interior point: the diagonal-Hessian variant (use_diag_hessian) and the
// inexact Newton-Krylov step (use_hvec_product): right-preconditioned GMRES on the KKT system with the
// quasi-Newton KKT matrix as preconditioner (reference computeKKTGMRESStep,
// src/ParOptInteriorPoint.cpp:5796-6191).  Each application of the preconditioner is the same fused
// bordered solve as the quasi-Newton step (one panel-dot pass, host algebra on the replicated
// (c+k)-sized blocks, one panel-axpy pass), with the right-hand side (bx, alpha * other blocks) of the
// alpha-scaled overload of solveKKTDiagSystem (:2441-2614); A px and Z^T px come from the weighted Gram
// matrix W, so the projections fproj / aproj cost one reduction pass.
#include <math.h>

#include <algorithm>

#include "ip.hpp"

namespace po {

int InteriorPoint::ensureHdiag() {
  if (hdiag) return PO_OK;
  hdiag = vec_new(ctx, n);  // zero-initialised, as the reference's hdiag before the first evaluation
  return hdiag ? PO_OK : PO_ERR_HIP;
}

// y = [K0 + quasi-Newton low-rank]^-1 (bx, alpha * b_other): px (and pzl, pzu, the step minima when
// `full`); the dense blocks of `out` carry the low-rank correction only when `full`, exactly as the
// reference corrects only step.x inside the GMRES loop (:5936-5950) and the whole step at the end
// (:6142-6160).
int InteriorPoint::solveKKTAlpha(const double *bx, double alpha, const Dense &b, double mu, bool use_qn,
                                 bool full, double tau, Dense &out) {
  const double beta_mu = options.real("rel_bound_barrier") * mu;
  int k = 0;
  std::vector<const double *> P = panel(use_qn, &k);
  if (k != wk) {
    set_error("internal: panel width changed between setUpKKTSystem and solve (%d vs %d)", k, wk);
    return PO_ERR_ARG;
  }
  const int m = c + k;
  PO_TRY(k_d1s(ctx, bounds(), bx, Dinv->d, alpha, beta_mu, n, tvec->d));
  std::vector<double> dots(m > 0 ? m : 1, 0.0);
  if (m > 0) PO_TRY(k_mdot(ctx, tvec->d, P.data(), m, n, dots.data()));
  std::vector<double> yz(c > 0 ? c : 1, 0.0), yz2(c > 0 ? c : 1, 0.0), zeta(k > 0 ? k : 1, 0.0);
  for (int i = 0; i < c; i++) {
    yz[i] = alpha * (b.z[i] + (b.zs[i] + vars.s[i] * b.s[i]) / vars.zs[i] -
                     (b.zt[i] + vars.t[i] * b.t[i]) / vars.zt[i]) -
            dots[i];
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
  std::vector<double> coef(m > 0 ? m : 1, 0.0);
  for (int i = 0; i < c; i++) coef[i] = yz[i] - yz2[i];
  for (int j = 0; j < k; j++) coef[c + j] = -zeta[j];
  ptpx.assign(m > 0 ? m : 1, 0.0);
  for (int i = 0; i < m; i++) {
    double v = dots[i];
    for (int j = 0; j < m; j++) v += W[i + (size_t)m * j] * coef[j];
    ptpx[i] = v;
  }
  ptpx_valid = true;
  merit_cache_valid = false;  // the step is about to change
  px_amax_valid = false;
  fused_merit_valid = false;
  w_comp_valid = w_merit_cache_valid = false;
  tdots_valid = false;
  residual_fused = false;
  vA_valid = false;
  PO_TRY(k_solve2s(ctx, bounds(), tvec->d, Dinv->d, coef.data(), P.data(), m, alpha, beta_mu, full ? 1 : 0, tau,
                   n, px->d, pzl->d, pzu->d, step_mins));
  for (int i = 0; i < c; i++) {
    const double zs1 = yz[i] - alpha * b.s[i];
    const double zt1 = -alpha * b.t[i] - yz[i];
    const double y2 = full ? yz2[i] : 0.0;
    out.z[i] = yz[i] - y2;
    out.zs[i] = zs1 - y2;
    out.zt[i] = zt1 + y2;
    out.s[i] = (alpha * b.zs[i] - vars.s[i] * zs1) / vars.zs[i] + (vars.s[i] * y2) / vars.zs[i];
    out.t[i] = (alpha * b.zt[i] - vars.t[i] * zt1) / vars.zt[i] - (vars.t[i] * y2) / vars.zt[i];
  }
  return PO_OK;
}

// The same solve with sparse constraints: K0^-1 is the quasi-definite apply (block or CSR form), the
// right-hand side gains d2 = alpha (bzw + (bzsw + sw bsw)/zsw - (bztw + tw btw)/ztw) (:2482-2508) and the step
// gains the five w blocks (:2557-2585).  Inside the GMRES loop (`full` false) the reference corrects only
// step.x for the quasi-Newton part, so the w blocks come from K0^-1 (d1 + Ac yz, d2) - one more apply.
int InteriorPoint::solveKKTAlphaW(const double *bx, double alpha, const Dense &b, double mu, bool use_qn,
                                  bool full, double tau, Dense &out) {
  const double beta_mu = options.real("rel_bound_barrier") * mu;
  int k = 0;
  std::vector<const double *> P = panel(use_qn, &k);
  if (k != wk) {
    set_error("internal: panel width changed between setUpKKTSystem and solve (%d vs %d)", k, wk);
    return PO_ERR_ARG;
  }
  const int m = c + k;
  WVars ws{wscalev[0]->d, wscalev[1]->d, wscalev[2]->d, wscalev[3]->d, wscalev[4]->d};
  PO_TRY(k_d1s(ctx, bounds(), bx, nullptr, alpha, beta_mu, n, d1v->d));
  PO_TRY(k_w_scale5(ctx, ws, wr(), alpha, nw));
  PO_TRY(k_w_d2(ctx, wv(), ws, nw, wd2->d));
  PO_TRY(applyK0(d1v->d, wd2->d, tvec, wyw));
  std::vector<double> dots(m > 0 ? m : 1, 0.0);
  if (m > 0) PO_TRY(k_mdot(ctx, tvec->d, P.data(), m, n, dots.data()));
  std::vector<double> yz(c > 0 ? c : 1, 0.0), yz2(c > 0 ? c : 1, 0.0), zeta(k > 0 ? k : 1, 0.0);
  for (int i = 0; i < c; i++) {
    yz[i] = alpha * (b.z[i] + (b.zs[i] + vars.s[i] * b.s[i]) / vars.zs[i] -
                     (b.zt[i] + vars.t[i] * b.t[i]) / vars.zt[i]) -
            dots[i];
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
  std::vector<double> coef(m > 0 ? m : 1, 0.0);
  for (int i = 0; i < c; i++) coef[i] = yz[i] - yz2[i];
  for (int j = 0; j < k; j++) coef[c + j] = -zeta[j];
  ptpx.assign(m > 0 ? m : 1, 0.0);
  for (int i = 0; i < m; i++) {
    double v = dots[i];
    for (int j = 0; j < m; j++) v += W[i + (size_t)m * j] * coef[j];
    ptpx[i] = v;
  }
  ptpx_valid = true;
  merit_cache_valid = false;
  px_amax_valid = false;
  fused_merit_valid = false;
  w_comp_valid = w_merit_cache_valid = false;
  tdots_valid = false;
  residual_fused = false;
  vA_valid = false;
  double mins_w[2] = {1.0, 1.0};
  if (!full && k > 0) {
    // w blocks without the quasi-Newton correction: K0^-1 (d1 + Ac yz, d2)
    if (c > 0) {
      PO_TRY(k_panel_axpy(ctx, xt->d, 1.0, d1v->d, 0.0, yz.data(), P.data(), c, n));
      PO_TRY(applyK0(xt->d, wd2->d, tvec, wyw));
    }
    PO_TRY(k_w_step(ctx, wv(), ws, wyw->d, 0, tau, wp(), nw, mins_w));
    PO_TRY(k_panel_axpy(ctx, d1v->d, 0.0, nullptr, 1.0, coef.data(), P.data(), m, n));
    PO_TRY(applyK0(d1v->d, wd2->d, tvec, wtmp2));
  } else {
    if (m > 0) PO_TRY(k_panel_axpy(ctx, d1v->d, 0.0, nullptr, 1.0, coef.data(), P.data(), m, n));
    PO_TRY(applyK0(d1v->d, wd2->d, tvec, wyw));
    PO_TRY(k_w_step(ctx, wv(), ws, wyw->d, 0, tau, wp(), nw, mins_w));
  }
  double mins_x[2] = {1.0, 1.0};
  PO_TRY(k_solve2s(ctx, bounds(), tvec->d, Dinv->d, coef.data(), P.data(), 0, alpha, beta_mu, full ? 1 : 0, tau,
                   n, px->d, pzl->d, pzu->d, mins_x));
  step_mins[0] = std::min(mins_x[0], mins_w[0]);
  step_mins[1] = std::min(mins_x[1], mins_w[1]);
  for (int i = 0; i < c; i++) {
    const double zs1 = yz[i] - alpha * b.s[i];
    const double zt1 = -alpha * b.t[i] - yz[i];
    const double y2 = full ? yz2[i] : 0.0;
    out.z[i] = yz[i] - y2;
    out.zs[i] = zs1 - y2;
    out.zt[i] = zt1 + y2;
    out.s[i] = (alpha * b.zs[i] - vars.s[i] * zs1) / vars.zs[i] + (vars.s[i] * y2) / vars.zs[i];
    out.t[i] = (alpha * b.zt[i] - vars.t[i] * zt1) / vars.zt[i] - (vars.t[i] * y2) / vars.zt[i];
  }
  return PO_OK;
}

// evalObjBarrierDeriv :5669-5766 for the step (px, p.s, p.t): the merit derivative without the penalty
int InteriorPoint::evalObjBarrierDeriv(const Dense &p, double *pmerit_) {
  const double beta = options.real("rel_bound_barrier");
  double out[6];
  PO_TRY(k_merit0(ctx, bounds(), px->d, 1.0, g->d, n, out));
  double ppos = out[2] * beta, pneg = out[3] * beta;
  if (has_w) {
    // the sparse slacks (:5711-5730, :5767) and, in the same pass, sum (cw - sw + tw)(Aw px - psw + ptw)
    // for the constraint-descent test of the Krylov loop
    const double *cw = nullptr;
    PO_TRY(sparseConAtIterate(&cw));
    PO_TRY(k_fill(ctx, wtmp2->d, nw, 0.0));
    if (prob->addSparseJacobian(1.0, x, px, wtmp2) != 0) return PO_ERR_USER;
    PO_TRY(k_w_merit(ctx, wv(), wp(), 1.0, gsw->d, gtw->d, cw, wtmp2->d, nw, w_merit_last));
    ppos += w_merit_last[2];
    pneg += w_merit_last[3];
  }
  for (int i = 0; i < c; i++) {
    if (p.s[i] > 0.0) ppos += p.s[i] / vars.s[i]; else pneg += p.s[i] / vars.s[i];
    if (p.t[i] > 0.0) ppos += p.t[i] / vars.t[i]; else pneg += p.t[i] / vars.t[i];
  }
  double pmerit = out[4] - barrier_param * (ppos + pneg);
  for (int i = 0; i < c; i++) pmerit += gamma_s[i] * p.s[i] + gamma_t[i] * p.t[i];
  if (has_w) pmerit += w_merit_last[6] + w_merit_last[7];
  *pmerit_ = pmerit;
  return PO_OK;
}

// The small least-squares problem of the Krylov step, on the host: the upper Hessenberg matrix of the Arnoldi
// relation as full columns (entry (r, col), r <= col + 1), reduced to triangular form column by column with plane
// rotations that are also applied to the right-hand side; solve() back-substitutes the leading block.
struct KrylovLeastSquares {
  int ld;
  std::vector<double> hcol, rot_c, rot_s, rhs;
  explicit KrylovLeastSquares(int msub)
      : ld(msub + 1), hcol((size_t)(msub + 1) * msub, 0.0), rot_c(msub, 0.0), rot_s(msub, 0.0), rhs(msub + 1, 0.0) {}
  double &h(int r, int col) { return hcol[(size_t)col * ld + r]; }
  // column `col` is complete (rows 0 .. col + 1): apply the earlier rotations, make and apply its own
  void closeColumn(int col) {
    for (int r = 0; r < col; r++) {
      const double a = h(r, col), b = h(r + 1, col);
      h(r, col) = a * rot_c[r] + b * rot_s[r];
      h(r + 1, col) = -a * rot_s[r] + b * rot_c[r];
    }
    const double a = h(col, col), b = h(col + 1, col);
    const double len = sqrt(a * a + b * b);
    rot_c[col] = a / len;
    rot_s[col] = b / len;
    h(col, col) = a * rot_c[col] + b * rot_s[col];
    h(col + 1, col) = -a * rot_s[col] + b * rot_c[col];
    const double g = rhs[col];
    rhs[col] = g * rot_c[col];
    rhs[col + 1] = -g * rot_s[col];
  }
  // out[0..k) = R^-1 rhs[0..k) for the leading k columns (out may be rhs itself)
  void solve(int k, double *out) {
    for (int r = k - 1; r >= 0; r--) {
      double v = rhs[r];
      for (int col = r + 1; col < k; col++) v = v - h(r, col) * out[col];
      out[r] = v / h(r, r);
    }
  }
};

int InteriorPoint::computeKKTGMRESStep(double rtol, double atol, bool use_qn, double tau, int *gmres_iters) {
  const int msub = options.integer("gmres_subspace_size");
  *gmres_iters = 0;
  if (msub <= 0) {
    if (ctx->rank == 0) fprintf(stderr, "ParOpt error: gmres_subspace_size not set\n");
    return PO_OK;
  }
  while ((int)gmresW.size() < msub + 1) {
    Vec *w = vec_new(ctx, n);
    if (!w) return PO_ERR_HIP;
    gmresW.push_back(w);
  }
  std::vector<Vec *> &Wk = gmresW;
  const double mu = barrier_param;
  if (has_w) PO_TRY(computeResidualW(mu));
  denseResidual(mu, res);
  KrylovLeastSquares ls(msub);
  std::vector<double> &gres = ls.rhs;
  std::vector<double> alpha(msub + 1, 0.0), y(msub, 0.0), fproj(msub, 0.0), aproj(msub, 0.0);
  // |b|: the x block from the residual pass, the bound blocks from its sums, the dense blocks here
  double beta = 0.0;
  for (int i = 0; i < c; i++) {
    beta += res.z[i] * res.z[i] + res.s[i] * res.s[i] + res.t[i] * res.t[i] + res.zs[i] * res.zs[i] +
            res.zt[i] * res.zt[i];
  }
  if (use_lower) beta += l2_rzl;
  if (use_upper) beta += l2_rzu;
  double cwinfeas = 0.0, cwscale = 0.0;
  if (has_w) {
    double sq[5];
    PO_TRY(k_w_sumsq5(ctx, wr(), nw, sq));
    beta += sq[0] + sq[1] + sq[2] + sq[3] + sq[4];
    cwinfeas = sqrt(sq[0]);  // |rzw| = |cw - sw + tw| (:5884-5890)
    if (cwinfeas != 0.0) cwscale = 1.0 / cwinfeas;
  }
  const double bnorm = sqrt(l2_rx + beta);
  beta *= 1.0 / (bnorm * bnorm);
  double cinfeas = 0.0, cscale = 0.0;
  for (int i = 0; i < c; i++) cinfeas += (cvals[i] - vars.s[i] + vars.t[i]) * (cvals[i] - vars.s[i] + vars.t[i]);
  if (cinfeas != 0.0) {
    cinfeas = sqrt(cinfeas);
    cscale = 1.0 / cinfeas;
  }
  gres[0] = bnorm;
  PO_TRY(k_panel_axpy(ctx, Wk[0]->d, 1.0 / gres[0], rx->d, 0.0, nullptr, nullptr, 0, n));
  alpha[0] = 1.0;
  int niters = 0;
  Dense p;
  p.resize(c);
  const int kq = (qn && use_qn) ? qn->size() : 0;
  std::vector<const double *> Z;
  if (kq > 0) Z = qn->zPointers();
  for (int i = 0; i < msub; i++) {
    if (has_w) {
      PO_TRY(solveKKTAlphaW(Wk[i]->d, alpha[i] / bnorm, res, mu, use_qn, false, tau, p));
    } else {
      PO_TRY(solveKKTAlpha(Wk[i]->d, alpha[i] / bnorm, res, mu, use_qn, false, tau, p));
    }
    PO_TRY(evalObjBarrierDeriv(p, &fproj[i]));
    aproj[i] = 0.0;
    for (int j = 0; j < c; j++) aproj[i] -= cscale * res.z[j] * (ptpx[j] - p.s[j] + p.t[j]);
    // sparse part (:5963-5973): -cwscale rzw.(Aw px) + cwscale rzw.(psw - ptw) with rzw = -(cw - sw + tw)
    if (has_w) aproj[i] += cwscale * w_merit_last[9];
    // W_{i+1} = H px - B px + W_i
    if (prob->evalHvecProduct(x, vars.z.data(), has_w ? wvar[0] : nullptr, px, Wk[i + 1]) != 0) {
      set_error("evalHvecProduct failed or is not provided by the problem");
      return PO_ERR_USER;
    }
    nhvec++;
    {
      std::vector<double> cf(kq + 1, 0.0);
      std::vector<const double *> V;
      V.push_back(px->d);
      if (qn && use_qn) {
        cf[0] = -qn->diag();
        if (kq > 0) {
          std::vector<double> rz(ptpx.begin() + c, ptpx.begin() + c + kq);
          qn->applyCompactInverse(rz.data());
          for (int j = 0; j < kq; j++) cf[1 + j] = rz[j];
          V.insert(V.end(), Z.begin(), Z.end());
        }
      }
      PO_TRY(k_panel_axpy(ctx, Wk[i + 1]->d, 1.0, Wk[i]->d, 1.0, cf.data(), V.data(), (int)V.size(), n));
    }
    alpha[i + 1] = alpha[i];
    for (int j = i; j >= 0; j--) {  // modified Gram-Schmidt against the basis, newest vector first (as :5986-5996)
      double d = 0.0;
      PO_TRY(k_reduce1(ctx, RED_DOT, Wk[i + 1]->d, Wk[j]->d, n, &d));
      const double hj = d + beta * alpha[i + 1] * alpha[j];
      ls.h(j, i) = hj;
      PO_TRY(k_axpy(ctx, Wk[i + 1]->d, -hj, Wk[j]->d, n));
      alpha[i + 1] -= hj * alpha[j];
    }
    double nrm2 = 0.0;
    PO_TRY(k_reduce1(ctx, RED_SUMSQ, Wk[i + 1]->d, nullptr, n, &nrm2));
    const double hlast = sqrt(nrm2 + beta * alpha[i + 1] * alpha[i + 1]);
    ls.h(i + 1, i) = hlast;
    PO_TRY(k_scale(ctx, Wk[i + 1]->d, n, 1.0 / hlast));
    alpha[i + 1] *= 1.0 / hlast;
    ls.closeColumn(i);
    niters++;
    ls.solve(niters, y.data());  // the current solution's coefficients (the rotated right-hand side is left alone)
    double fpr = 0.0, cpr = 0.0;
    for (int j = 0; j < niters; j++) {
      fpr += y[j] * fproj[j];
      cpr += y[j] * aproj[j];
    }
    const bool constraint_descent = cpr <= -0.01 * (cinfeas + cwinfeas);
    if (fpr < 0.0 || constraint_descent) {
      if (fabs(gres[i + 1]) < atol || fabs(gres[i + 1]) < rtol * bnorm) break;
    }
  }
  ls.solve(niters, gres.data());
  // u_x = sum gres_i W_i ; gamma scales every other block of the right-hand side
  double gamma = gres[0] * alpha[0];
  {
    std::vector<const double *> V;
    std::vector<double> cf;
    for (int i = 1; i < niters; i++) {
      V.push_back(Wk[i]->d);
      cf.push_back(gres[i]);
      gamma += gres[i] * alpha[i];
    }
    PO_TRY(k_panel_axpy(ctx, Wk[0]->d, 0.0, nullptr, gres[0], cf.data(), V.data(), (int)V.size(), n));
  }
  gamma /= bnorm;
  if (has_w) {
    PO_TRY(solveKKTAlphaW(Wk[0]->d, gamma, res, mu, use_qn, true, tau, step));
  } else {
    PO_TRY(solveKKTAlpha(Wk[0]->d, gamma, res, mu, use_qn, true, tau, step));
  }
  sx = sz = 1.0;
  double fpr = 0.0, cpr = 0.0;
  PO_TRY(evalObjBarrierDeriv(step, &fpr));
  for (int i = 0; i < c; i++) {
    const double deriv = ptpx[i] - step.s[i] + step.t[i];
    cpr += cscale * (cvals[i] - vars.s[i] + vars.t[i]) * deriv;
  }
  if (has_w) {
    // the reference's final test subtracts BOTH slack terms (:6171-6173): r.(Aw px) - r.psw - r.ptw with
    // r = cw - sw + tw, i.e. the loop's sum minus 2 r.ptw; wresv[0] still holds -r
    double rptw = 0.0;
    PO_TRY(k_reduce1(ctx, RED_DOT, wresv[0]->d, wstepv[2]->d, nw, &rptw));
    cpr += cwscale * (w_merit_last[9] + 2.0 * rptw);
  }
  *gmres_iters = (fpr < 0.0 || cpr < -0.01 * (cinfeas + cwinfeas)) ? niters : -niters;
  return PO_OK;
}

}  // namespace po
