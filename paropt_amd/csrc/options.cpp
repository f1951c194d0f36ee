// ParOptOptions registry of the interior-point / trust-region / MMA drivers (host only): names, defaults and
// ranges of src/ParOptInteriorPoint.cpp:536-727, src/ParOptTrustRegion.cpp:739-847, src/ParOptMMA.cpp:234-289;
// unknown names, wrong types and out-of-range values are refused (src/ParOptOptions.cpp:310-386).
#include "ip.hpp"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

namespace po {

// ================================================================================================
// Options
// ================================================================================================
Options::Options() {
  auto S = [&](const char *n, const char *v) {
    Entry x;
    x.type = STR;
    x.s = v ? v : "";
    e[n] = x;
  };
  auto F = [&](const char *n, double v, double lo, double hi) {
    Entry x;
    x.type = FLOAT;
    x.f = v;
    x.flo = lo;
    x.fhi = hi;
    e[n] = x;
  };
  auto B = [&](const char *n, int v) {
    Entry x;
    x.type = BOOL;
    x.i = v;
    x.ilo = 0;
    x.ihi = 1;
    e[n] = x;
  };
  auto I = [&](const char *n, int v, int lo, int hi) {
    Entry x;
    x.type = INT;
    x.i = v;
    x.ilo = lo;
    x.ihi = hi;
    e[n] = x;
  };
  auto E = [&](const char *n, const char *v, std::vector<std::string> ch) {
    Entry x;
    x.type = ENUM;
    x.s = v;
    x.choices = ch;
    e[n] = x;
  };
  // names / defaults / ranges: src/ParOptInteriorPoint.cpp:536-727
  S("output_file", "paropt.out");
  S("problem_name", "");
  F("max_bound_value", 1e20, 0.0, 1e300);
  F("abs_res_tol", 1e-6, 0.0, 1e20);
  F("rel_func_tol", 0.0, 0.0, 1e20);
  F("abs_step_tol", 0.0, 0.0, 1e20);
  F("init_barrier_param", 0.1, 0.0, 1e20);
  F("penalty_gamma", 1000.0, 0.0, 1e20);
  F("penalty_descent_fraction", 0.3, 1e-6, 1.0);
  F("min_rho_penalty_search", 0.0, 0.0, 1e20);
  F("init_rho_penalty_search", 0.0, 0.0, 1e20);
  F("armijo_constant", 1e-5, 0.0, 1.0);
  F("monotone_barrier_fraction", 0.25, 0.0, 1.0);
  F("monotone_barrier_power", 1.1, 1.0, 10.0);
  F("rel_bound_barrier", 1.0, 0.0, 1e20);
  F("min_fraction_to_boundary", 0.95, 0.0, 1.0);
  F("qn_sigma", 0.0, 0.0, 1e20);
  F("nk_switch_tol", 1e-3, 0.0, 1e20);
  F("eisenstat_walker_alpha", 1.5, 0.0, 2.0);
  F("eisenstat_walker_gamma", 1.0, 0.0, 1.0);
  F("max_gmres_rtol", 0.1, 0.0, 1.0);
  F("gmres_atol", 1e-30, 0.0, 1.0);
  F("function_precision", 1e-10, 0.0, 1.0);
  F("design_precision", 1e-14, 0.0, 1.0);
  F("start_affine_multiplier_min", 1.0, 0.0, 1e20);
  F("gradient_check_step_length", 1e-6, 0.0, 1.0);
  B("use_line_search", 1);
  B("use_backtracking_alpha", 0);
  B("sequential_linear_method", 0);
  B("use_quasi_newton_update", 1);
  B("use_hvec_product", 0);
  B("use_diag_hessian", 0);
  B("use_qn_gmres_precon", 1);
  I("qn_subspace_size", 10, 0, 1000);
  I("max_major_iters", 5000, 0, 1000000);
  I("max_line_iters", 10, 1, 100);
  I("iterative_refinement_steps", 1, 0, 10);
  I("gmres_subspace_size", 0, 0, 1000);
  I("write_output_frequency", 10, 0, 1000000);
  I("step_verification_frequency", -1, -1000000, 1000000);
  I("gradient_verification_frequency", -1, -1000000, 1000000);
  I("hessian_reset_freq", 1000000, 1, 1000000);
  I("output_level", 0, 0, 1000000);
  E("qn_type", "bfgs", {"bfgs", "scaled_bfgs", "sr1", "none"});
  E("qn_update_type", "skip_negative_curvature", {"skip_negative_curvature", "damped_update"});
  E("qn_diag_type", "yty_over_yts",
    {"yty_over_yts", "yts_over_sts", "inner_yty_over_yts", "inner_yts_over_sts"});
  E("norm_type", "infinity", {"infinity", "l1", "l2"});
  E("barrier_strategy", "monotone",
    {"monotone", "mehrotra", "mehrotra_predictor_corrector", "complementarity_fraction"});
  E("starting_point_strategy", "affine_step",
    {"least_squares_multipliers", "affine_step", "no_start_strategy"});
}

void Options::addTrustRegionDefaults() {
  auto S = [&](const char *n, const char *v) {
    Entry x;
    x.type = STR;
    x.s = v ? v : "";
    e[n] = x;
  };
  auto F = [&](const char *n, double v, double lo, double hi) {
    Entry x;
    x.type = FLOAT;
    x.f = v;
    x.flo = lo;
    x.fhi = hi;
    e[n] = x;
  };
  auto B = [&](const char *n, int v) {
    Entry x;
    x.type = BOOL;
    x.i = v;
    x.ilo = 0;
    x.ihi = 1;
    e[n] = x;
  };
  auto I = [&](const char *n, int v, int lo, int hi) {
    Entry x;
    x.type = INT;
    x.i = v;
    x.ilo = lo;
    x.ihi = hi;
    e[n] = x;
  };
  auto E = [&](const char *n, const char *v, std::vector<std::string> ch) {
    Entry x;
    x.type = ENUM;
    x.s = v;
    x.choices = ch;
    e[n] = x;
  };
  S("tr_output_file", "paropt.tr");
  F("tr_init_size", 0.1, 0.0, 1e20);
  F("tr_min_size", 1e-3, 0.0, 1e20);
  F("tr_max_size", 1.0, 0.0, 1e20);
  F("tr_eta", 0.25, 0.0, 1.0);
  F("tr_bound_relax", 1e-4, 0.0, 1e20);
  I("tr_write_output_frequency", 10, 0, 1000000);
  B("tr_adaptive_gamma_update", 1);
  E("tr_accept_step_strategy", "penalty_method", {"penalty_method", "filter_method"});
  B("filter_sufficient_reduction", 1);
  F("filter_gamma", 1e-5, 0.0, 1.0);
  B("filter_has_feas_restore_phase", 1);
  B("tr_use_soc", 0);
  B("tr_soc_update_qn", 0);
  I("tr_max_soc_iterations", 20, 0, 1000000);
  I("tr_max_iterations", 200, 0, 1000000);
  F("tr_l1_tol", 1e-6, 0.0, 1e20);
  F("tr_linfty_tol", 1e-6, 0.0, 1e20);
  F("tr_infeas_tol", 1e-5, 0.0, 1e20);
  F("tr_penalty_gamma_max", 1e4, 0.0, 1e20);
  F("tr_penalty_gamma_min", 0.0, 0.0, 1e20);
  E("tr_adaptive_objective", "linear_objective",
    {"constant_objective", "linear_objective", "subproblem_objective"});
  E("tr_adaptive_constraint", "linear_constraint", {"linear_constraint", "subproblem_constraint"});
  E("tr_steering_barrier_strategy", "mehrotra_predictor_corrector",
    {"monotone", "mehrotra", "mehrotra_predictor_corrector", "complementarity_fraction", "default"});
  E("tr_steering_starting_point_strategy", "affine_step",
    {"least_squares_multipliers", "affine_step", "no_start_strategy", "default"});
}

void Options::addMMADefaults() {
  auto F = [&](const char *n, double v, double lo, double hi) {
    Entry x;
    x.type = FLOAT;
    x.f = v;
    x.flo = lo;
    x.fhi = hi;
    e[n] = x;
  };
  Entry s;
  s.type = STR;
  s.s = "paropt.mma";
  e["mma_output_file"] = s;
  Entry it;
  it.type = INT;
  it.i = 200;
  it.ilo = 0;
  it.ihi = 1000000;
  e["mma_max_iterations"] = it;
  Entry lin;
  lin.type = BOOL;
  lin.i = 0;
  lin.ilo = 0;
  lin.ihi = 1;
  e["mma_use_constraint_linearization"] = lin;
  F("mma_l1_tol", 1e-6, 0.0, 1e20);
  F("mma_linfty_tol", 1e-6, 0.0, 1e20);
  F("mma_infeas_tol", 1e-5, 0.0, 1e20);
  F("mma_asymptote_contract", 0.7, 0.0, 1.0);
  F("mma_asymptote_relax", 1.2, 1.0, 1e20);
  F("mma_init_asymptote_offset", 0.5, 0.0, 1.0);
  F("mma_min_asymptote_offset", 0.01, 0.0, 1e20);
  F("mma_max_asymptote_offset", 10.0, 0.0, 1e20);
  F("mma_bound_relax", 0.0, 0.0, 1e20);
  F("mma_eps_regularization", 1e-5, 0.0, 1e20);
  F("mma_delta_regularization", 1e-3, 0.0, 1e20);
  F("mma_move_limit", 0.2, 0.0, 1e20);
}

int Options::set(const char *name, const char *value) {
  auto it = e.find(name);
  if (it == e.end()) {
    set_error("ParOptOptions: unknown option %s", name);
    return PO_ERR_OPTION;
  }
  Entry &x = it->second;
  if (x.type == STR) {
    x.s = value ? value : "";
    return PO_OK;
  }
  if (x.type == ENUM) {
    for (const std::string &ch : x.choices) {
      if (value && ch == value) {
        x.s = value;
        return PO_OK;
      }
    }
    set_error("ParOptOptions: %s is not a value of enum option %s", value ? value : "(null)", name);
    return PO_ERR_OPTION;
  }
  set_error("ParOptOptions: option %s is not a string/enum option", name);
  return PO_ERR_OPTION;
}
int Options::set(const char *name, int value) {
  auto it = e.find(name);
  if (it == e.end()) {
    set_error("ParOptOptions: unknown option %s", name);
    return PO_ERR_OPTION;
  }
  Entry &x = it->second;
  if (x.type == BOOL) {
    x.i = value ? 1 : 0;
    return PO_OK;
  }
  if (x.type == INT) {
    if (value < x.ilo || value > x.ihi) {
      set_error("ParOptOptions: %d out of range [%d, %d] for %s", value, x.ilo, x.ihi, name);
      return PO_ERR_OPTION;
    }
    x.i = value;
    return PO_OK;
  }
  set_error("ParOptOptions: option %s is not an int/bool option", name);
  return PO_ERR_OPTION;
}
int Options::set(const char *name, double value) {
  auto it = e.find(name);
  if (it == e.end()) {
    set_error("ParOptOptions: unknown option %s", name);
    return PO_ERR_OPTION;
  }
  Entry &x = it->second;
  if (x.type != FLOAT) {
    set_error("ParOptOptions: option %s is not a float option", name);
    return PO_ERR_OPTION;
  }
  if (value < x.flo || value > x.fhi) {
    set_error("ParOptOptions: %g out of range [%g, %g] for %s", value, x.flo, x.fhi, name);
    return PO_ERR_OPTION;
  }
  x.f = value;
  return PO_OK;
}
int Options::visit(const Options *skip, po_option_visitor fn, void *user) const {
  for (const auto &kv : e) {
    if (skip && skip->has(kv.first.c_str())) continue;
    const Entry &x = kv.second;
    std::vector<const char *> ch;
    for (const std::string &c : x.choices) ch.push_back(c.c_str());
    fn(user, kv.first.c_str(), (int)x.type, x.s.c_str(), x.i, x.ilo, x.ihi, x.f, x.flo, x.fhi, (int)ch.size(),
       ch.empty() ? nullptr : ch.data());
  }
  return PO_OK;
}
void Options::adoptExtras(const Options &src, const Options &base) {
  for (const auto &kv : src.e)
    if (!base.has(kv.first.c_str())) e[kv.first] = kv.second;
}
const char *Options::str(const char *name) const { return e.at(name).s.c_str(); }
int Options::integer(const char *name) const { return e.at(name).i; }
double Options::real(const char *name) const { return e.at(name).f; }

}  // namespace po
