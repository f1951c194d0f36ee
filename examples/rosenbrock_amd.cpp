// The reference's C++ usage pattern (examples/rosenbrock/rosenbrock.cpp: subclass ParOptProblem,
// fill host arrays through getArray, hand it to ParOptInteriorPoint) on the MI355X path through
// include/ParOptAMD.hpp: f = sum (1-x_i)^2 + 100 (x_{i+1}-x_i^2)^2,
// c0 = 0.25 - sum x^2 >= 0, c1 = 10 + sum_{i even} x_i >= 0, -2 <= x <= 1, x0 = -1, and with nwcon=5
// the reference example's sparse constraints cw_i = 1 - sum_{k<5} x[1 + 6 i + k] >= 0
// (rosenbrock.cpp:131-184: nwcon = 5, nw = 5, nwstart = 1, nwskip = 1).
//
// build: g++ -std=c++17 -Iinclude examples/rosenbrock_amd.cpp -Lparopt_amd -lparopt_amd
//        -Wl,-rpath,$PWD/paropt_amd -o examples/rosenbrock_amd ; run: ./examples/rosenbrock_amd nvars=100
#include <stdlib.h>
#include <string.h>

#include "ParOptAMD.hpp"

class Rosenbrock : public ParOptProblem {
 public:
  Rosenbrock(po_ctx ctx, int n, int _nwcon = 0) : ParOptProblem(ctx) {
    nw = 5;
    nwstart = 1;
    nwskip = 1;
    setProblemSizes(n, 2, _nwcon);
    setNumInequalities(2, _nwcon);
  }
  void evalSparseCon(ParOptVec *x, ParOptVec *out) {
    ParOptScalar *xvals, *cw;
    x->getArray(&xvals);
    out->getArray(&cw);
    for (int i = 0, j = nwstart; i < nwcon; i++, j += nw + nwskip) {
      cw[i] = 1.0;
      for (int k = 0; k < nw; k++) cw[i] -= xvals[j + k];
    }
  }
  void addSparseJacobian(ParOptScalar alpha, ParOptVec *, ParOptVec *px, ParOptVec *out) {
    ParOptScalar *pxvals, *cw;
    px->getArray(&pxvals);
    out->getArray(&cw);
    for (int i = 0, j = nwstart; i < nwcon; i++, j += nw + nwskip) {
      for (int k = 0; k < nw; k++) cw[i] -= alpha * pxvals[j + k];
    }
  }
  void addSparseJacobianTranspose(ParOptScalar alpha, ParOptVec *, ParOptVec *pzw, ParOptVec *out) {
    ParOptScalar *outvals, *zw;
    out->getArray(&outvals);
    pzw->getArray(&zw);
    for (int i = 0, j = nwstart; i < nwcon; i++, j += nw + nwskip) {
      for (int k = 0; k < nw; k++) outvals[j + k] -= alpha * zw[i];
    }
  }
  void addSparseInnerProduct(ParOptScalar alpha, ParOptVec *, ParOptVec *cvec, ParOptScalar *A) {
    ParOptScalar *cvals;
    cvec->getArray(&cvals);
    for (int i = 0, j = nwstart; i < nwcon; i++, j += nw + nwskip) {
      for (int k = 0; k < nw; k++) A[i] += alpha * cvals[j + k];
    }
  }
  int nw, nwstart, nwskip;
  void getVarsAndBounds(ParOptVec *xvec, ParOptVec *lbvec, ParOptVec *ubvec) {
    ParOptScalar *x, *lb, *ub;
    xvec->getArray(&x);
    lbvec->getArray(&lb);
    ubvec->getArray(&ub);
    for (int i = 0; i < nvars; i++) {
      x[i] = -1.0;
      lb[i] = -2.0;
      ub[i] = 1.0;
    }
  }
  int evalObjCon(ParOptVec *xvec, ParOptScalar *fobj, ParOptScalar *cons) {
    ParOptScalar *x;
    xvec->getArray(&x);
    double f = 0.0, c0 = 0.25, c1 = 10.0;
    for (int i = 0; i + 1 < nvars; i++) {
      const double r = x[i + 1] - x[i] * x[i];
      f += (1.0 - x[i]) * (1.0 - x[i]) + 100.0 * r * r;
    }
    for (int i = 0; i < nvars; i++) c0 -= x[i] * x[i];
    for (int i = 0; i < nvars; i += 2) c1 += x[i];
    *fobj = f;
    cons[0] = c0;
    cons[1] = c1;
    return 0;
  }
  int evalObjConGradient(ParOptVec *xvec, ParOptVec *gvec, ParOptVec **Ac) {
    ParOptScalar *x, *g, *a0, *a1;
    xvec->getArray(&x);
    gvec->getArray(&g);
    Ac[0]->getArray(&a0);
    Ac[1]->getArray(&a1);
    for (int i = 0; i < nvars; i++) g[i] = 0.0;
    for (int i = 0; i + 1 < nvars; i++) {
      const double r = x[i + 1] - x[i] * x[i];
      g[i] += -2.0 * (1.0 - x[i]) - 400.0 * r * x[i];
      g[i + 1] += 200.0 * r;
    }
    for (int i = 0; i < nvars; i++) a0[i] = -2.0 * x[i];
    for (int i = 0; i < nvars; i += 2) a1[i] = 1.0;  // odd entries stay zero, as in the reference
    return 0;
  }
};

int main(int argc, char *argv[]) {
  int nvars = 100, nwcon = 0, use_tr = 0;
  for (int k = 1; k < argc; k++) {
    sscanf(argv[k], "nvars=%d", &nvars);
    sscanf(argv[k], "nwcon=%d", &nwcon);
    if (strcmp(argv[k], "algorithm=tr") == 0) use_tr = 1;
  }
  po_ctx ctx = NULL;
  if (po_ctx_create(0, &ctx) != 0) {
    fprintf(stderr, "no MI355X available: %s\n", po_last_error());
    return 2;
  }
  Rosenbrock *rosen = new Rosenbrock(ctx, nvars, nwcon);
  rosen->incref();
  ParOptOptions *options = new ParOptOptions();
  options->incref();
  options->setOption("qn_type", "bfgs");
  options->setOption("qn_subspace_size", 10);
  options->setOption("abs_res_tol", 1e-6);
  options->setOption("barrier_strategy", "monotone");
  options->setOption("max_major_iters", 150);
  if (use_tr) {
    // the reference's generic entry point (src/ParOptOptimizer.cpp:108-183): trust region over the
    // quadratic subproblem with the interior-point method as sub-solver
    options->setOption("algorithm", "tr");
    options->setOption("tr_max_iterations", 80);
    options->setOption("tr_output_file", "");
    ParOptOptimizer *optimizer = new ParOptOptimizer(rosen, options);
    optimizer->incref();
    optimizer->optimize();
    ParOptVec *xt;
    ParOptScalar *zt;
    optimizer->getOptimizedPoint(&xt, &zt, NULL, NULL, NULL);
    ParOptScalar ft, ct[2];
    rosen->evalObjCon(xt, &ft, ct);
    int ntr = 0;
    for (const char *h = optimizer->getTrustRegionHistory(); *h; h++) ntr += (*h == '\n');
    printf("{\"algorithm\": \"tr\", \"fobj\": %.15e, \"xnorm\": %.15e, \"z0\": %.15e, \"z1\": %.15e, "
           "\"table_lines\": %d}\n", ft, xt->norm(), zt[0], zt[1], ntr);
    optimizer->decref();
    options->decref();
    rosen->decref();
    po_ctx_destroy(ctx);
    return 0;
  }
  ParOptInteriorPoint *opt = new ParOptInteriorPoint(rosen, options);
  opt->incref();
  int rc = opt->optimize();
  int niter, neval, ngeval;
  opt->getIterationCounters(&niter, &neval, &ngeval);
  ParOptVec *x, *zw = NULL;
  ParOptScalar *z;
  opt->getOptimizedPoint(&x, &z, &zw, NULL, NULL);
  ParOptScalar fobj, cons[2];
  rosen->evalObjCon(x, &fobj, cons);
  printf("{\"rc\": %d, \"niter\": %d, \"neval\": %d, \"ngeval\": %d, \"fobj\": %.15e, \"xnorm\": %.15e, "
         "\"z0\": %.15e, \"z1\": %.15e, \"zwnorm\": %.15e}\n", rc, niter, neval, ngeval, fobj, x->norm(),
         z[0], z[1], zw ? zw->norm() : 0.0);
  opt->decref();
  options->decref();
  rosen->decref();
  po_ctx_destroy(ctx);
  return rc;
}
