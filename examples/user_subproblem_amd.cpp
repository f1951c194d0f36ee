// A trust-region subproblem WRITTEN BY THE USER under ParOptTrustRegion: the extension point of the reference
// (ParOptTrustRegionSubproblem, src/ParOptTrustRegion.h:15-151; ParOptOptimizer::setTrustRegionSubproblem,
// src/ParOptOptimizer.cpp:226-237) on include/ParOptAMD.hpp.  The subclass below restates the reference's quadratic
// model (src/ParOptTrustRegion.cpp:27-466: f(s) = fk + gk.s + 1/2 s.B s, c(s) = ck + Ak s, box |s| <= tr inside the
// variable bounds, quasi-Newton update from the Lagrangian gradient difference) with nothing but the public vector and
// quasi-Newton classes; the device solver calls its virtuals through the callback table of po_trsub_create_callbacks.
// On the separable quadratic of the goldens it reproduces the compiled reference's iteration table
// (tests/golden/tr_quadratic_n200_c3_bfgs.npz; tests/test_cpp_facade.py).
//
// build: make -C examples user_subproblem_amd
// run:   ./examples/user_subproblem_amd n=200 c=3 [driver=objects|optimizer] [opt.<name>=<value> ...]
#include <stdlib.h>
#include <string.h>

#include <string>
#include <vector>

#include "separable_quadratic.hpp"

class UserQuadraticSubproblem : public ParOptTrustRegionSubproblem {
 public:
  UserQuadraticSubproblem(ParOptProblem *_prob, ParOptCompactQuasiNewton *_qn)
      : ParOptTrustRegionSubproblem(_prob->getMPIComm()), prob(_prob), qn(_qn), fk(0.0), ft(0.0), update_type(0) {
    prob->incref();
    if (qn) qn->incref();
    prob->getProblemSizes(&n, &m, NULL);
    int nineq = 0;
    prob->getNumInequalities(&nineq, NULL);
    setProblemSizes(n, m, 0);
    setNumInequalities(nineq, 0);
    ParOptVec **all[] = {&xk, &lb, &ub, &lk, &uk, &gk, &gt, &t, &xt};
    for (ParOptVec **v : all) {
      *v = prob->createDesignVec();
      (*v)->incref();
    }
    for (int i = 0; i < 2 * m; i++) {
      ParOptVec *a = prob->createDesignVec();
      a->incref();
      (i < m ? Ak : At).push_back(a);
    }
    ck.assign(m > 0 ? m : 1, 0.0);
    ct.assign(m > 0 ? m : 1, 0.0);
  }
  ~UserQuadraticSubproblem() {
    ParOptVec *all[] = {xk, lb, ub, lk, uk, gk, gt, t, xt};
    for (ParOptVec *v : all) v->decref();
    for (ParOptVec *v : Ak) v->decref();
    for (ParOptVec *v : At) v->decref();
    if (qn) qn->decref();
    prob->decref();
  }
  // ---- ParOptTrustRegionSubproblem ----
  ParOptCompactQuasiNewton *getQuasiNewton() { return qn; }
  void initModelAndBounds(double tr_size) {  // :141-151
    prob->getVarsAndBounds(xk, lb, ub);
    setTrustRegionBounds(tr_size);
    prob->evalObjCon(xk, &fk, ck.data());
    prob->evalObjConGradient(xk, gk, Ak.data());
  }
  void setTrustRegionBounds(double tr_size) {  // :153-173 (bounds on the step)
    ParOptScalar *x, *l, *u, *sl, *su;
    xk->getArray(&x);
    lb->getArray(&l);
    ub->getArray(&u);
    lk->getArray(&sl);
    uk->getArray(&su);
    for (int i = 0; i < n; i++) {
      sl[i] = l[i] - x[i] > -tr_size ? l[i] - x[i] : -tr_size;
      su[i] = u[i] - x[i] < tr_size ? u[i] - x[i] : tr_size;
    }
  }
  int evalTrialStepAndUpdate(int update_flag, ParOptVec *step, ParOptScalar *z, ParOptVec *zw, ParOptScalar *fobj,
                             ParOptScalar *cons) {  // :175-212
    xt->copyValues(xk);
    xt->axpy(1.0, step);
    int fail = prob->evalObjCon(xt, &ft, ct.data());
    fail = fail || prob->evalObjConGradient(xt, gt, At.data());
    *fobj = ft;
    for (int i = 0; i < m; i++) cons[i] = ct[i];
    if (qn && update_flag) {  // y = [gt - At^T z] - [gk - Ak^T z]
      t->copyValues(gt);
      t->axpy(-1.0, gk);
      for (int i = 0; i < m; i++) {
        t->axpy(-z[i], At[i]);
        t->axpy(z[i], Ak[i]);
      }
      update_type = qn->update(xt, z, zw, step, t);
    }
    return fail;
  }
  int acceptTrialStep(ParOptVec *step, ParOptScalar *, ParOptVec *) {  // :214-224
    xk->axpy(1.0, step);
    fk = ft;
    std::swap(gk, gt);
    for (int i = 0; i < m; i++) {
      ck[i] = ct[i];
      std::swap(Ak[i], At[i]);
    }
    return 0;
  }
  void rejectTrialStep() {
    ft = 0.0;
    for (int i = 0; i < m; i++) ct[i] = 0.0;
  }
  int getQuasiNewtonUpdateType() { return update_type; }
  int getLinearModel(ParOptVec **_xk = NULL, ParOptScalar *_fk = NULL, ParOptVec **_gk = NULL,
                     const ParOptScalar **_ck = NULL, ParOptVec ***_Ak = NULL, ParOptVec **_lb = NULL,
                     ParOptVec **_ub = NULL) {
    if (_xk) *_xk = xk;
    if (_fk) *_fk = fk;
    if (_gk) *_gk = gk;
    if (_ck) *_ck = ck.data();
    if (_Ak) *_Ak = Ak.data();
    if (_lb) *_lb = lb;
    if (_ub) *_ub = ub;
    return m;
  }
  // ---- the model as the interior point's problem (functions of the step) ----
  void getVarsAndBounds(ParOptVec *step, ParOptVec *l, ParOptVec *u) {  // :278-285
    step->copyValues(lk);
    step->axpy(1.0, uk);
    step->scale(0.5);
    l->copyValues(lk);
    u->copyValues(uk);
  }
  int evalObjCon(ParOptVec *step, ParOptScalar *fobj, ParOptScalar *cons) {  // :290-323
    if (!step) {
      *fobj = fk;
      for (int i = 0; i < m; i++) cons[i] = ck[i];
      return 0;
    }
    *fobj = fk + gk->dot(step);
    if (qn) {
      qn->mult(step, t);
      *fobj += 0.5 * step->dot(t);
    }
    if (m > 0) step->mdot(Ak.data(), m, cons);
    for (int i = 0; i < m; i++) cons[i] += ck[i];
    return 0;
  }
  int evalObjConGradient(ParOptVec *step, ParOptVec *g, ParOptVec **Ac) {  // :328-343
    g->copyValues(gk);
    if (qn) qn->multAdd(1.0, step, g);
    for (int i = 0; Ac && i < m; i++) Ac[i]->copyValues(Ak[i]);
    return 0;
  }

 private:
  ParOptProblem *prob;
  ParOptCompactQuasiNewton *qn;
  int n, m;
  ParOptVec *xk, *lb, *ub, *lk, *uk, *gk, *gt, *t, *xt;
  std::vector<ParOptVec *> Ak, At;
  ParOptScalar fk, ft;
  std::vector<ParOptScalar> ck, ct;
  int update_type;
};

int main(int argc, char *argv[]) {
  int n = 200, m = 3;
  long seed = 0;
  std::string driver = "objects";
  std::vector<std::pair<std::string, std::string>> extra;
  for (int k = 1; k < argc; k++) {
    sscanf(argv[k], "n=%d", &n);
    sscanf(argv[k], "c=%d", &m);
    sscanf(argv[k], "seed=%ld", &seed);
    if (strncmp(argv[k], "driver=", 7) == 0) driver = argv[k] + 7;
    if (strncmp(argv[k], "opt.", 4) == 0) {
      const char *eq = strchr(argv[k], '=');
      if (eq) extra.emplace_back(std::string(argv[k] + 4, (size_t)(eq - argv[k] - 4)), std::string(eq + 1));
    }
  }
  po_ctx ctx = NULL;
  if (po_ctx_create(0, &ctx) != 0) {
    fprintf(stderr, "no MI355X available: %s\n", po_last_error());
    return 2;
  }
  SeparableQuadratic *problem = new SeparableQuadratic(ctx, n, m, (uint64_t)seed);
  problem->incref();
  ParOptOptions *options = new ParOptOptions();
  options->incref();
  ParOptOptimizer::addDefaultOptions(options);
  options->setOption("algorithm", "tr");
  options->setOption("output_file", "");
  options->setOption("tr_output_file", "");
  for (auto &kv : extra) {
    const int type = options->getOptionType(kv.first.c_str());
    if (type == ParOptOptions::PAROPT_FLOAT_OPTION) {
      options->setOption(kv.first.c_str(), atof(kv.second.c_str()));
    } else if (type == ParOptOptions::PAROPT_INT_OPTION || type == ParOptOptions::PAROPT_BOOLEAN_OPTION) {
      options->setOption(kv.first.c_str(), atoi(kv.second.c_str()));
    } else {
      options->setOption(kv.first.c_str(), kv.second.c_str());
    }
  }
  ParOptLBFGS *qn = new ParOptLBFGS(problem, options->getIntOption("qn_subspace_size"));
  qn->incref();
  UserQuadraticSubproblem *subproblem = new UserQuadraticSubproblem(problem, qn);
  subproblem->incref();

  ParOptVec *x = NULL;
  std::string table;
  ParOptOptimizer *optimizer = NULL;
  ParOptInteriorPoint *ip = NULL;
  ParOptTrustRegion *tr = NULL;
  if (driver == "infeas") {
    // ParOptInfeasSubproblem (src/ParOptTrustRegion.h:293-374) built by hand over the user-written subproblem and over
    // the library's ParOptQuadraticSubproblem on the same problem: the steering LP of the first trust-region iteration
    // (linear objective scaled by 0.5, linearised constraints, sequential linear method), one solve each
    ParOptQuadraticSubproblem *libsub = new ParOptQuadraticSubproblem(problem, qn);
    libsub->incref();
    ParOptTrustRegionSubproblem *subs[2] = {subproblem, libsub};
    options->setOption("sequential_linear_method", 1);
    options->setOption("use_line_search", 0);
    for (int k = 0; k < 2; k++) {
      subs[k]->initModelAndBounds(options->getFloatOption("tr_init_size"));
      ParOptInfeasSubproblem *infeas = new ParOptInfeasSubproblem(
          subs[k], ParOptInfeasSubproblem::PAROPT_LINEAR_OBJECTIVE, ParOptInfeasSubproblem::PAROPT_LINEAR_CONSTRAINT);
      infeas->incref();
      infeas->setObjectiveScaling(0.5);
      ParOptInteriorPoint *solver = new ParOptInteriorPoint(infeas, options);
      solver->incref();
      solver->optimize();
      ParOptVec *step = NULL;
      ParOptScalar *z = NULL;
      solver->getOptimizedPoint(&step, &z, NULL, NULL, NULL);
      ParOptScalar f = 0.0;
      std::vector<ParOptScalar> con(m > 0 ? m : 1, 0.0);
      infeas->evalObjCon(step, &f, con.data());
      printf("infeas %s: fobj %.15e |step| %.15e maxabs %.15e z", k == 0 ? "user" : "library", f, step->norm(),
             step->maxabs());
      for (int i = 0; i < m; i++) printf(" %.15e", z[i]);
      printf(" con");
      for (int i = 0; i < m; i++) printf(" %.15e", con[i]);
      printf("\n");
      solver->decref();
      infeas->decref();
    }
    libsub->decref();
    subproblem->decref();
    qn->decref();
    options->decref();
    problem->decref();
    po_ctx_destroy(ctx);
    return 0;
  }
  if (driver == "optimizer") {
    optimizer = new ParOptOptimizer(problem, options);
    optimizer->incref();
    optimizer->setTrustRegionSubproblem(subproblem);
    optimizer->optimize();
    optimizer->getOptimizedPoint(&x, NULL, NULL, NULL, NULL);
    table = optimizer->getTrustRegionHistory();
  } else {
    ip = new ParOptInteriorPoint(subproblem, options);
    ip->incref();
    tr = new ParOptTrustRegion(subproblem, options);
    tr->incref();
    tr->optimize(ip);
    tr->getOptimizedPoint(&x);
    table = tr->getHistory();
  }
  fputs(table.c_str(), stdout);
  ParOptScalar fobj = 0.0;
  std::vector<ParOptScalar> cons(m > 0 ? m : 1, 0.0);
  problem->evalObjCon(x, &fobj, cons.data());
  printf("\nfinal: fobj %.15e  |x| %.15e\n", fobj, x->norm());
  if (optimizer) optimizer->decref();
  if (tr) tr->decref();
  if (ip) ip->decref();
  subproblem->decref();
  qn->decref();
  options->decref();
  problem->decref();
  po_ctx_destroy(ctx);
  return 0;
}
