// The reference's examples/rosenbrock/sparse_rosenbrock.cpp usage pattern on the MI355X path: subclass
// ParOptSparseProblem, declare the CSR pattern of the sparse Jacobian once, fill the constraint values and
// the Jacobian entries in evalSparseObjCon / evalSparseObjConGradient.  Same problem as the reference
// example: f = sum (1-x_i)^2 + 100 (x_{i+1}-x_i^2)^2, c0 = 0.25 - sum x^2 >= 0, c1 = 10 + sum_{i even} x_i
// >= 0, and nvars-1 overlapping sparse constraints cw_i = 1 - x_i^2 - x_{i+1}^2 >= 0; -2 <= x <= 1, x0 = -1.
// The Schur complement of the sparse block is tridiagonal here; it is assembled, factored (nested-dissection
// ordered, level-scheduled sparse Cholesky) and solved on the GPU.
//
// build: g++ -std=c++17 -Iinclude examples/sparse_rosenbrock_amd.cpp -Lparopt_amd -lparopt_amd
//        -Wl,-rpath,$PWD/paropt_amd -o examples/sparse_rosenbrock_amd ; run: ./examples/sparse_rosenbrock_amd nvars=100
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "ParOptAMD.hpp"

class SparseRosenbrock : public ParOptSparseProblem {
 public:
  SparseRosenbrock(po_ctx ctx, int n) : ParOptSparseProblem(ctx) {
    setProblemSizes(n, 2, n - 1);
    setNumInequalities(2, n - 1);
    std::vector<int> rowp(n), cols(2 * (n - 1));
    for (int i = 0; i < n - 1; i++) {
      rowp[i] = 2 * i;
      cols[2 * i] = i;
      cols[2 * i + 1] = i + 1;
    }
    rowp[n - 1] = 2 * (n - 1);
    setSparseJacobianData(rowp.data(), cols.data());
  }
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
  int evalSparseObjCon(ParOptVec *xvec, ParOptScalar *fobj, ParOptScalar *cons, ParOptVec *sparse) {
    ParOptScalar *x, *c;
    xvec->getArray(&x);
    sparse->getArray(&c);
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
    for (int i = 0; i < nwcon; i++) c[i] = 1.0 - x[i] * x[i] - x[i + 1] * x[i + 1];
    return 0;
  }
  int evalSparseObjConGradient(ParOptVec *xvec, ParOptVec *gvec, ParOptVec **Ac, ParOptScalar *data) {
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
    for (int i = 0; i < nvars; i += 2) a1[i] = 1.0;
    for (int i = 0; i < nwcon; i++) {
      data[2 * i] = -2.0 * x[i];
      data[2 * i + 1] = -2.0 * x[i + 1];
    }
    return 0;
  }
};

int main(int argc, char *argv[]) {
  int nvars = 100;
  for (int k = 1; k < argc; k++) sscanf(argv[k], "nvars=%d", &nvars);
  po_ctx ctx = NULL;
  if (po_ctx_create(0, &ctx) != 0) {
    fprintf(stderr, "no MI355X available: %s\n", po_last_error());
    return 2;
  }
  SparseRosenbrock *rosen = new SparseRosenbrock(ctx, nvars);
  rosen->incref();
  ParOptOptions *options = new ParOptOptions();
  options->incref();
  options->setOption("qn_type", "bfgs");
  options->setOption("qn_subspace_size", 10);
  options->setOption("abs_res_tol", 1e-6);
  options->setOption("barrier_strategy", "monotone");
  options->setOption("max_major_iters", 150);
  options->setOption("output_file", "");
  ParOptInteriorPoint *opt = new ParOptInteriorPoint(rosen, options);
  opt->incref();
  int rc = opt->optimize();
  int niter, neval, ngeval;
  opt->getIterationCounters(&niter, &neval, &ngeval);
  ParOptVec *x, *zw = NULL;
  ParOptScalar *z;
  opt->getOptimizedPoint(&x, &z, &zw, NULL, NULL);
  ParOptScalar fobj, cons[2];
  ParOptVec *cw = new ParOptBasicVec(ctx, nvars - 1);
  cw->incref();
  rosen->evalSparseObjCon(x, &fobj, cons, cw);
  const char *info = rosen->getFactorInfo();
  printf("{\"rc\": %d, \"niter\": %d, \"neval\": %d, \"ngeval\": %d, \"fobj\": %.15e, \"xnorm\": %.15e, "
         "\"z0\": %.15e, \"z1\": %.15e, \"zwnorm\": %.15e, \"factor_info\": \"%s\"}\n", rc, niter, neval, ngeval, fobj,
         x->norm(), z[0], z[1], zw ? zw->norm() : 0.0, info ? info : "");
  cw->decref();
  opt->decref();
  options->decref();
  rosen->decref();
  po_ctx_destroy(ctx);
  return rc;
}
