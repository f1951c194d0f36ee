/* The flat C ABI from plain C (C99): the built-in separable quadratic workload solved by the
 * interior point, then by the trust-region driver and by MMA -- the three algorithms of the
 * reference's ParOptOptimizer -- without any C++ or Python in between.
 *
 * build: gcc -std=c99 -Iinclude examples/c_abi_quadratic.c -Lparopt_amd -lparopt_amd \
 *            -Wl,-rpath,$PWD/paropt_amd -Wl,-rpath-link,/opt/rocm/lib -o examples/c_abi_quadratic
 * run:   ./examples/c_abi_quadratic [n]
 */
#include <stdio.h>
#include <stdlib.h>

#include "paropt_amd.h"

#define CHECK(call)                                                          \
  do {                                                                       \
    int _rc = (call);                                                        \
    if (_rc != 0) {                                                          \
      fprintf(stderr, "%s failed (%d): %s\n", #call, _rc, po_last_error()); \
      return 2;                                                              \
    }                                                                        \
  } while (0)

int main(int argc, char **argv) {
  long n = argc > 1 ? atol(argv[1]) : 100000;
  po_ctx ctx = NULL;
  if (po_ctx_create(0, &ctx) != 0) {
    fprintf(stderr, "no MI355X available: %s\n", po_last_error());
    return 2;
  }
  po_problem prob = NULL;
  CHECK(po_problem_create_separable(ctx, PO_PROBLEM_QUADRATIC, n, 3, 0, 1.0, 100.0, &prob));

  /* ParOptInteriorPoint */
  po_ip ip = NULL;
  CHECK(po_ip_create(prob, &ip));
  CHECK(po_ip_set_option_str(ip, "output_file", ""));
  CHECK(po_ip_set_option_float(ip, "abs_res_tol", 1e-8));
  CHECK(po_ip_set_option_float(ip, "start_affine_multiplier_min", 0.01));
  CHECK(po_ip_optimize(ip, NULL));
  int niter = 0, neval = 0, ngeval = 0;
  double f_ip = 0.0, rho = 0.0;
  CHECK(po_ip_get_counters(ip, &niter, &neval, &ngeval));
  CHECK(po_ip_get_objective(ip, &f_ip, &rho));

  /* ParOptTrustRegion over the quadratic subproblem */
  po_tr tr = NULL;
  CHECK(po_tr_create(prob, &tr));
  CHECK(po_tr_set_option_str(tr, "tr_output_file", ""));
  CHECK(po_tr_set_option_int(tr, "tr_max_iterations", 80));
  CHECK(po_tr_set_option_float(tr, "tr_max_size", 2.0));
  CHECK(po_tr_optimize(tr));
  int tr_iters = 0;
  double f_tr = 0.0;
  CHECK(po_tr_get_state(tr, NULL, &tr_iters, NULL, NULL, NULL, &f_tr, NULL));

  /* ParOptMMA */
  po_mma mma = NULL;
  CHECK(po_mma_create(prob, &mma));
  CHECK(po_mma_set_option_str(mma, "mma_output_file", ""));
  CHECK(po_mma_set_option_int(mma, "mma_max_iterations", 40));
  CHECK(po_mma_optimize(mma));
  int mma_iters = 0, sub_iters = 0;
  double f_mma = 0.0;
  CHECK(po_mma_get_state(mma, &mma_iters, &sub_iters, &f_mma, NULL));

  printf("{\"n\": %ld, \"ip\": {\"niter\": %d, \"fobj\": %.12e}, \"tr\": {\"iters\": %d, \"fobj\": %.12e}, "
         "\"mma\": {\"iters\": %d, \"sub_iters\": %d, \"fobj\": %.12e}}\n",
         n, niter, f_ip, tr_iters, f_tr, mma_iters, sub_iters, f_mma);
  po_mma_destroy(mma);
  po_tr_destroy(tr);
  po_ip_destroy(ip);
  po_problem_destroy(prob);
  po_ctx_destroy(ctx);
  return 0;
}
