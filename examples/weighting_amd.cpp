// BASELINE.json's config 4 (topology-optimisation style: a convex objective, a few dense constraints and ONE sparse
// "weighting" constraint per group of consecutive variables) written the way a ParOpt USER writes it: a
// ParOptSparseProblem subclass against the reference's C++ interface (src/ParOptProblem.h:301-395; the reference's own
// instances of the pattern are examples/rosenbrock/rosenbrock.cpp:131-184 and examples/dmo_truss/
// dmo_truss_analysis.py:650-679), here on include/ParOptAMD.hpp.  Compiled OUTSIDE libparopt_amd.so; the evaluations are
// the user's own HIP kernels on the device arrays behind ParOptVec.
//
//   f(x)    = sum_i b_i^2 / (eps + x_i)                     (examples/random_convex/random_convex.py:44-51,66, Q = I)
//   c_j(x)  = beta_j - a_j . x >= 0,  beta_j = 0.25 sum_i a_ji
//   cw_i(x) = 1 - sum_{k < nw} x[i nw + k] >= 0,  i < nwcon  (one constraint per group of nw consecutive variables)
//   0 <= x <= 1
//
// The sparse Jacobian goes to the library through the reference's own API and nothing else:
// setSparseJacobianData(rowp, cols) with the CSR pattern of the groups, evalSparseObjCon for f, c, cw and
// evalSparseObjConGradient for g, Ac and the nnz Jacobian entries (all -1).  The library RECOGNISES the pattern (rows
// of equal length over consecutive columns at equal spacing, every entry the same value -- checked on the device after
// each gradient evaluation) and runs its fused group kernels; any other pattern or non-uniform entries take the general
// CSR path.  The only facade extension used is the device form of the gradient callback (the entries are written
// into the library's device array by a kernel instead of into a host array that is then copied: 160 MB per call at
// 1 M constraints x 20 variables).
//
// build (see examples/Makefile):  libweighting_user.so (bench.py --boundary facade --nwcon ..., tests) and weighting_amd
//   ./examples/weighting_amd n=2000000 c=4 nwcon=100000 nw=20 iters=30
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ParOptAMD.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCon = 64;
constexpr double kEps = 1e-3;

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__global__ void __launch_bounds__(kThreads)
    objective_kernel(const double *__restrict__ b, const double *__restrict__ x, long n, double *__restrict__ partials) {
  __shared__ double sm[kThreads / 64];
  double acc = 0.0;
  const long npairs = n >> 1;
  for (long q = (long)blockIdx.x * kThreads + threadIdx.x; q < npairs; q += (long)gridDim.x * kThreads) {
    const double2 bv = reinterpret_cast<const double2 *>(b)[q], xv = reinterpret_cast<const double2 *>(x)[q];
    acc += bv.x * bv.x / (kEps + xv.x);
    acc += bv.y * bv.y / (kEps + xv.y);
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) acc += b[n - 1] * b[n - 1] / (kEps + x[n - 1]);
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) partials[blockIdx.x] = (sm[0] + sm[1]) + (sm[2] + sm[3]);
}
__global__ void __launch_bounds__(64) objective_final_kernel(const double *__restrict__ partials, int nblocks,
                                                             double *__restrict__ out) {
  double acc = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 64) acc += partials[i];
  acc = wave_sum(acc);
  if (threadIdx.x == 0) out[0] = acc;
}
__global__ void __launch_bounds__(kThreads)
    gradient_kernel(const double *__restrict__ b, const double *__restrict__ x, long n, double *__restrict__ g) {
  const long npairs = n >> 1;
  for (long q = (long)blockIdx.x * kThreads + threadIdx.x; q < npairs; q += (long)gridDim.x * kThreads) {
    const double2 bv = reinterpret_cast<const double2 *>(b)[q], xv = reinterpret_cast<const double2 *>(x)[q];
    const double d0 = kEps + xv.x, d1 = kEps + xv.y;
    reinterpret_cast<double2 *>(g)[q] = make_double2(-(bv.x * bv.x) / (d0 * d0), -(bv.y * bv.y) / (d1 * d1));
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) {
    const double d0 = kEps + x[n - 1];
    g[n - 1] = -(b[n - 1] * b[n - 1]) / (d0 * d0);
  }
}
struct JacTables {
  const double *src[kMaxCon];
  double *dst[kMaxCon];
};
__global__ void __launch_bounds__(kThreads) jacobian_kernel(JacTables t, int ncon, long n) {
  const long npairs = n >> 1;
  for (long q = (long)blockIdx.x * kThreads + threadIdx.x; q < npairs; q += (long)gridDim.x * kThreads) {
    for (int j = 0; j < ncon; j++) {
      const double2 v = reinterpret_cast<const double2 *>(t.src[j])[q];
      reinterpret_cast<double2 *>(t.dst[j])[q] = make_double2(-v.x, -v.y);
    }
  }
  if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0)
    for (int j = 0; j < ncon; j++) t.dst[j][n - 1] = -t.src[j][n - 1];
}
// cw_i = 1 - (x[i nw] + ... + x[i nw + nw - 1]): 16 lanes per group, consecutive lanes on consecutive variables
// (the sum runs over the variables in ascending order within a lane stride, then across the 16 lanes)
__global__ void __launch_bounds__(kThreads)
    weighting_kernel(const double *__restrict__ x, long nwcon, int nw, double *__restrict__ cw) {
  const int sub = threadIdx.x & 15;
  for (long i = ((long)blockIdx.x * kThreads + threadIdx.x) >> 4; i < nwcon; i += ((long)gridDim.x * kThreads) >> 4) {
    double acc = 0.0;
    for (int k = sub; k < nw; k += 16) acc += x[i * nw + k];
    for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 16);
    if (sub == 0) cw[i] = 1.0 - acc;
  }
}
__global__ void __launch_bounds__(kThreads) fill_kernel(double *__restrict__ y, long n, double v) {
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) y[i] = v;
}

double *device_array(ParOptVec *v) {
  double *d = NULL;
  po_vec_get_device_array(v->handle(), &d);
  return d;
}

}  // namespace

class WeightingConvex : public ParOptSparseProblem {
 public:
  // nglobal design variables and nwcon_global weighting constraints over the ranks of the context (contiguous row
  // blocks; a group never straddles two ranks: the shard size must be a multiple of nw), ncon dense inequalities
  WeightingConvex(po_ctx _ctx, int64_t nglobal, int _ncon, uint64_t _seed, int64_t nwcon_global, int _nw)
      : ParOptSparseProblem(_ctx), seed(_seed), nw(_nw), b(NULL), d_partials(NULL), d_f(NULL), h_f(NULL), ok(true),
        n_obj_evals(0), n_grad_evals(0) {
    int rank = 0, size = 1;
    po_ctx_rank(_ctx, &rank, &size);
    const int64_t base = nglobal / size, rem = nglobal % size;
    nlocal = base + (rank < rem ? 1 : 0);
    offset = rank * base + (rank < rem ? rank : rem);
    if (size > 1 && (nlocal % nw != 0 || offset % nw != 0)) {
      fprintf(stderr, "weighting_amd: with %d ranks the shard size must be a multiple of nw = %d\n", size, nw);
      ok = false;
    }
    // this rank's groups: group i lives where its first variable lives
    int64_t first = offset / nw, count = nlocal / nw;
    if (first > nwcon_global) first = nwcon_global;
    if (first + count > nwcon_global) count = nwcon_global - first;
    nwlocal = count;
    setProblemSizes((int)nlocal, _ncon, (int)nwlocal);
    setNumInequalities(_ncon, (int)nwlocal);
    // the CSR pattern of the sparse Jacobian, through the reference's interface (src/ParOptProblem.h:306-312)
    {
      std::vector<int> rowp((size_t)nwlocal + 1), cols((size_t)nwlocal * nw);
      for (int64_t i = 0; i <= nwlocal; i++) rowp[(size_t)i] = (int)(i * nw);
      for (int64_t j = 0; j < nwlocal * nw; j++) cols[(size_t)j] = (int)j;
      setSparseJacobianData(rowp.data(), cols.data());
    }
    stream = (hipStream_t)po_ctx_stream(_ctx);
    hipDeviceProp_t prop;
    int dev = 0;
    (void)hipGetDevice(&dev);
    (void)hipGetDeviceProperties(&prop, dev);
    grid = 4 * (prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256);
    const long blocks_needed = ((nlocal >> 1) + kThreads - 1) / kThreads;
    if (blocks_needed < grid) grid = blocks_needed > 0 ? (int)blocks_needed : 1;
    b = createDesignVec();
    b->incref();
    po_vec_fill_hash(b->handle(), seed, 2, offset, 1.0, 0.0);
    for (int j = 0; j < ncon; j++) {
      ParOptVec *a = createDesignVec();
      a->incref();
      po_vec_fill_hash(a->handle(), seed, 100 + j, offset, 1.0, 0.0);
      A.push_back(a);
    }
    beta.assign(ncon > 0 ? ncon : 1, 0.0);
    if (ncon > 0) {
      ParOptVec *ones = createDesignVec();
      ones->incref();
      ones->set(1.0);
      ones->mdot(A.data(), ncon, beta.data());
      ones->decref();
      for (int j = 0; j < ncon; j++) beta[j] *= 0.25;
    }
    (void)hipMalloc((void **)&d_partials, sizeof(double) * grid);
    (void)hipMalloc((void **)&d_f, sizeof(double));
    (void)hipHostMalloc((void **)&h_f, sizeof(double), hipHostMallocDefault);
  }
  ~WeightingConvex() {
    (void)hipStreamSynchronize(stream);
    if (b) b->decref();
    for (ParOptVec *a : A) a->decref();
    if (d_partials) (void)hipFree(d_partials);
    if (d_f) (void)hipFree(d_f);
    if (h_f) (void)hipHostFree(h_f);
  }

  void getVarsAndBounds(ParOptVec *x, ParOptVec *lb, ParOptVec *ub) {
    po_vec_fill_hash(x->handle(), seed, 3, offset, 0.9, 0.05);
    lb->set(0.0);
    ub->set(1.0);
  }

  // ParOptSparseProblem::evalSparseObjCon (src/ParOptProblem.h:330-331): f, c and the sparse constraint values
  int evalSparseObjCon(ParOptVec *x, ParOptScalar *fobj, ParOptScalar *cons, ParOptVec *sparse_con) {
    if (!ok) return 1;
    n_obj_evals++;
    const double *xd = device_array(x);
    objective_kernel<<<grid, kThreads, 0, stream>>>(device_array(b), xd, (long)nlocal, d_partials);
    objective_final_kernel<<<1, 64, 0, stream>>>(d_partials, grid, d_f);
    if (nwlocal > 0) {
      long blocks = (nwlocal * 16 + kThreads - 1) / kThreads;
      if (blocks > 8 * grid) blocks = 8 * grid;
      weighting_kernel<<<(int)blocks, kThreads, 0, stream>>>(xd, (long)nwlocal, nw, device_array(sparse_con));
    }
    (void)hipMemcpyAsync(h_f, d_f, sizeof(double), hipMemcpyDeviceToHost, stream);
    if (ncon > 0) {
      x->mdot(A.data(), ncon, cons);  // ParOptVec::mdot (collective, host result; its synchronisation covers h_f)
    } else {
      (void)hipStreamSynchronize(stream);
    }
    double f = *h_f;
    po_ctx_allreduce(ctx, &f, 1, 0);
    *fobj = f;
    for (int j = 0; j < ncon; j++) cons[j] = beta[j] - cons[j];
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }

  // the reference's form (src/ParOptProblem.h:333-334): entries into a HOST array -- kept complete, but the library
  // calls the device form below
  int evalSparseObjConGradient(ParOptVec *x, ParOptVec *g, ParOptVec **Ac, ParOptScalar *data) {
    const int fail = gradients(x, g, Ac);
    for (int64_t j = 0; j < nwlocal * nw; j++) data[j] = -1.0;
    return fail;
  }
  int evalSparseObjConGradientDevice(ParOptVec *x, ParOptVec *g, ParOptVec **Ac, ParOptScalar *device_data) {
    const int fail = gradients(x, g, Ac);
    const long nnz = (long)(nwlocal * nw);
    if (nnz > 0) fill_kernel<<<grid, kThreads, 0, stream>>>(device_data, nnz, -1.0);
    return fail != 0 || hipGetLastError() != hipSuccess;
  }

  // algorithmic HBM bytes of this problem's OWN kernels so far: objective 16 n + constraints 8 (n + w); gradient 24 n,
  // Jacobian rewrite 16 n ncon, Jacobian entries 8 nnz
  double ownKernelBytes(int jacobian_rewritten) const {
    const double n = (double)nlocal, w = (double)nwlocal;
    return 8.0 * ((2.0 * n + n + w) * n_obj_evals +
                  (3.0 * n + (jacobian_rewritten ? 2.0 * ncon * n : 0.0) + w * nw) * n_grad_evals);
  }

  int64_t nlocal, offset, nwlocal;
  uint64_t seed;
  int nw;

 private:
  int gradients(ParOptVec *x, ParOptVec *g, ParOptVec **Ac) {
    if (!ok) return 1;
    n_grad_evals++;
    gradient_kernel<<<grid, kThreads, 0, stream>>>(device_array(b), device_array(x), (long)nlocal, device_array(g));
    if (Ac && ncon > 0) {
      for (int j0 = 0; j0 < ncon; j0 += kMaxCon) {
        JacTables t;
        const int wdt = ncon - j0 < kMaxCon ? ncon - j0 : kMaxCon;
        for (int j = 0; j < kMaxCon; j++) {
          t.src[j] = j < wdt ? device_array(A[j0 + j]) : NULL;
          t.dst[j] = j < wdt ? device_array(Ac[j0 + j]) : NULL;
        }
        jacobian_kernel<<<grid, kThreads, 0, stream>>>(t, wdt, (long)nlocal);
      }
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }
  hipStream_t stream;
  int grid;
  ParOptVec *b;
  std::vector<ParOptVec *> A;
  std::vector<double> beta;
  double *d_partials, *d_f, *h_f;
  bool ok;
  long n_obj_evals, n_grad_evals;
};

// ---- C entry points for bench.py --boundary facade and the tests (ctypes) -------------------------------------------
extern "C" {
void *wt_problem_create_weighting(po_ctx ctx, int64_t nglobal, int ncon, uint64_t seed, int64_t nwcon, int nw) {
  WeightingConvex *p = new WeightingConvex(ctx, nglobal, ncon, seed, nwcon, nw);
  p->incref();
  return p;
}
void *wt_problem_create(po_ctx ctx, int64_t nglobal, int ncon, uint64_t seed) {
  return wt_problem_create_weighting(ctx, nglobal, ncon, seed, nglobal / 20, 20);
}
po_problem wt_problem_handle(void *p) { return static_cast<WeightingConvex *>(p)->handle(); }
void wt_problem_sizes(void *p, int64_t *nlocal, int64_t *offset) {
  *nlocal = static_cast<WeightingConvex *>(p)->nlocal;
  *offset = static_cast<WeightingConvex *>(p)->offset;
}
void wt_problem_set_linear_constraints(void *p, int flag) { static_cast<WeightingConvex *>(p)->setLinearConstraints(flag); }
void wt_problem_set_deferred_reductions(void *p, int flag) { static_cast<WeightingConvex *>(p)->setDeferredReductions(flag); }
double wt_problem_own_kernel_bytes(void *p, int jacobian_rewritten) {
  return static_cast<WeightingConvex *>(p)->ownKernelBytes(jacobian_rewritten);
}
void wt_problem_destroy(void *p) { static_cast<WeightingConvex *>(p)->decref(); }
}

#ifndef WEIGHTING_NO_MAIN
int main(int argc, char *argv[]) {
  long n = 2000000, nwcon = -1;
  int c = 4, iters = 30, nw = 20;
  for (int k = 1; k < argc; k++) {
    sscanf(argv[k], "n=%ld", &n);
    sscanf(argv[k], "c=%d", &c);
    sscanf(argv[k], "iters=%d", &iters);
    sscanf(argv[k], "nwcon=%ld", &nwcon);
    sscanf(argv[k], "nw=%d", &nw);
  }
  if (nwcon < 0) nwcon = n / nw;
  po_ctx ctx = NULL;
  if (po_ctx_create(0, &ctx) != 0) {
    fprintf(stderr, "no MI355X available: %s\n", po_last_error());
    return 2;
  }
  WeightingConvex *prob = new WeightingConvex(ctx, n, c, 0, nwcon, nw);
  prob->incref();
  ParOptOptions *options = new ParOptOptions();
  options->incref();
  options->setOption("algorithm", "ip");
  options->setOption("qn_type", "bfgs");
  options->setOption("qn_subspace_size", 10);
  options->setOption("abs_res_tol", 1e-8);
  options->setOption("starting_point_strategy", "affine_step");
  options->setOption("barrier_strategy", "monotone");
  options->setOption("start_affine_multiplier_min", 0.01);
  options->setOption("max_major_iters", iters);
  options->setOption("output_file", "");
  ParOptInteriorPoint *opt = new ParOptInteriorPoint(prob, options);
  opt->incref();
  int rc = opt->optimize();
  int niter, neval, ngeval;
  opt->getIterationCounters(&niter, &neval, &ngeval);
  ParOptVec *x;
  ParOptScalar *z;
  opt->getOptimizedPoint(&x, &z, NULL, NULL, NULL);
  printf("{\"rc\": %d, \"niter\": %d, \"neval\": %d, \"ngeval\": %d, \"xnorm\": %.15e, \"z0\": %.15e, \"factor\": \"%s\"}\n",
         rc, niter, neval, ngeval, x->norm(), c > 0 ? z[0] : 0.0, prob->getFactorInfo() ? prob->getFactorInfo() : "");
  opt->decref();
  options->decref();
  prob->decref();
  po_ctx_destroy(ctx);
  return rc;
}
#endif
