// BASELINE.json's metric workload (config 3) written the way a ParOpt USER writes a problem: a ParOptProblem
// subclass against the reference's C++ interface (src/ParOptProblem.h:42-296; the reference's own instance of
// this problem is examples/random_convex/random_convex.py:44-126, its C++ shape examples/rosenbrock/
// rosenbrock.cpp:49-191), here on include/ParOptAMD.hpp.  It is compiled OUTSIDE libparopt_amd.so and sees
// nothing of the library's internals: the evaluations are the user's own HIP kernels on the device arrays behind
// ParOptVec (po_vec_get_device_array), launched on the context's stream (po_ctx_stream), plus ParOptVec::mdot
// for the constraint products exactly as a reference C++ problem would call it.
//
//   f(x)   = sum_i b_i^2 / (eps + x_i)            (the Q = I, Affine = diag(eps) case of random_convex.py:44-51,66)
//   c_j(x) = beta_j - a_j . x >= 0,  beta_j = 0.25 sum_i a_ji   (:57,69,110-111)
//   0 <= x <= 1,  x0 = 0.05 + 0.9 u                (:33-35)
// b, a_j, u are counter-hash arrays (pure functions of seed, array id and GLOBAL index: any sharding sees the same
// data), the same arrays the library's built-in SeparableProblem("convex") uses, so that the two can be compared
// iteration by iteration (tests/test_gpu_user_problem.py).  Like the reference example, evalObjConGradient
// REWRITES the whole constraint Jacobian at every call (Ac[j] <- -a_j).
//
// build (see examples/Makefile):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -x hip -Iinclude examples/random_convex_amd.cpp \
//         -Lparopt_amd -lparopt_amd -Wl,-rpath,'$ORIGIN/../paropt_amd' -o examples/random_convex_amd
//   ./examples/random_convex_amd n=1000000 c=32 iters=30
// The same source built with -shared -DRANDOM_CONVEX_NO_MAIN is the library bench.py --boundary facade and the
// parity test load through the extern "C" block at the end.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "ParOptAMD.hpp"

namespace {

constexpr int kThreads = 256;
constexpr int kMaxCon = 128;
constexpr double kEps = 1e-3;  // random_convex.py:104

__device__ __forceinline__ double wave_sum(double v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// one partial sum of b^2 / (eps + x) per workgroup (16 bytes per lane and operand; the tail element alone)
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
// fixed-order sum of the workgroup partials (bit-reproducible for a fixed grid)
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

// Ac[j] <- -a_j for every constraint in one launch
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

double *device_array(ParOptVec *v) {
  double *d = NULL;
  po_vec_get_device_array(v->handle(), &d);
  return d;
}

}  // namespace

class RandomConvex : public ParOptProblem {
 public:
  // nglobal design variables over the ranks of the context (contiguous row blocks, as the reference's var_range,
  // src/ParOptInteriorPoint.cpp:213-229), ncon dense inequality constraints
  RandomConvex(po_ctx _ctx, int64_t nglobal, int _ncon, uint64_t _seed)
      : ParOptProblem(_ctx), seed(_seed), b(NULL), d_partials(NULL), d_f(NULL), h_f(NULL), cons_out(NULL),
        n_obj_evals(0), n_grad_evals(0) {
    int rank = 0, size = 1;
    po_ctx_rank(_ctx, &rank, &size);
    const int64_t base = nglobal / size, rem = nglobal % size;
    nlocal = base + (rank < rem ? 1 : 0);
    offset = rank * base + (rank < rem ? rank : rem);
    setProblemSizes((int)nlocal, _ncon, 0);
    setNumInequalities(_ncon, 0);
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
    if (ncon > 0) {  // beta_j = 0.25 sum_i a_ji: products with a vector of ones (collective)
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
  ~RandomConvex() {
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

  int evalObjCon(ParOptVec *x, ParOptScalar *fobj, ParOptScalar *cons) {
    n_obj_evals++;
    const double *xd = device_array(x), *bd = device_array(b);
    objective_kernel<<<grid, kThreads, 0, stream>>>(bd, xd, (long)nlocal, d_partials);
    objective_final_kernel<<<1, 64, 0, stream>>>(d_partials, grid, d_f);
    if (deferred_reductions) {
      // the library sums d_f over the ranks and fills *fobj / cons when the solver flushes its batch; the sign
      // and offset of the constraints are applied in the hook
      po_ctx_reduce_device(ctx, d_f, 1, 0, fobj);
      if (ncon > 0) x->mdot(A.data(), ncon, cons);
      cons_out = cons;
      po_ctx_after_reduce(ctx, &RandomConvex::finish_constraints, this);
      return 0;
    }
    // reference semantics: every value is final when the callback returns.  The copy of the objective part is
    // queued BEFORE the constraint products, whose host synchronisation then covers both.
    (void)hipMemcpyAsync(h_f, d_f, sizeof(double), hipMemcpyDeviceToHost, stream);
    if (ncon > 0) {
      x->mdot(A.data(), ncon, cons);  // ParOptVec::mdot, src/ParOptVec.cpp:152-170 (collective, host result)
    } else {
      (void)hipStreamSynchronize(stream);
    }
    double f = *h_f;
    po_ctx_allreduce(ctx, &f, 1, 0);  // MPI_Allreduce(SUM) of the rank-local parts; nothing on one rank
    *fobj = f;
    for (int j = 0; j < ncon; j++) cons[j] = beta[j] - cons[j];
    return 0;
  }

  int evalObjConGradient(ParOptVec *x, ParOptVec *g, ParOptVec **Ac) {
    n_grad_evals++;
    gradient_kernel<<<grid, kThreads, 0, stream>>>(device_array(b), device_array(x), (long)nlocal, device_array(g));
    if (Ac && ncon > 0) {  // (Ac == NULL only after setLinearConstraints(1))
      for (int j0 = 0; j0 < ncon; j0 += kMaxCon) {
        JacTables t;
        const int w = ncon - j0 < kMaxCon ? ncon - j0 : kMaxCon;
        for (int j = 0; j < kMaxCon; j++) {
          t.src[j] = j < w ? device_array(A[j0 + j]) : NULL;
          t.dst[j] = j < w ? device_array(Ac[j0 + j]) : NULL;
        }
        jacobian_kernel<<<grid, kThreads, 0, stream>>>(t, w, (long)nlocal);
      }
    }
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }

  // algorithmic HBM bytes of this problem's OWN kernels so far (the library cannot see them): 16 n per objective
  // evaluation, 24 n + 16 n ncon per gradient evaluation that rewrites the Jacobian
  double ownKernelBytes(int jacobian_rewritten) const {
    return 8.0 * (double)nlocal * (2.0 * n_obj_evals + (3.0 + (jacobian_rewritten ? 2.0 * ncon : 0.0)) * n_grad_evals);
  }

  int64_t nlocal, offset;
  uint64_t seed;

 private:
  static void finish_constraints(void *self) {
    RandomConvex *me = static_cast<RandomConvex *>(self);
    for (int j = 0; j < me->ncon; j++) me->cons_out[j] = me->beta[j] - me->cons_out[j];
  }
  hipStream_t stream;
  int grid;
  ParOptVec *b;
  std::vector<ParOptVec *> A;
  std::vector<double> beta;
  double *d_partials, *d_f, *h_f, *cons_out;
  long n_obj_evals, n_grad_evals;
};

// ---- C entry points for bench.py --boundary facade and tests/test_gpu_user_problem.py (ctypes) ---------------------
extern "C" {
void *rc_problem_create(po_ctx ctx, int64_t nglobal, int ncon, uint64_t seed) {
  RandomConvex *p = new RandomConvex(ctx, nglobal, ncon, seed);
  p->incref();
  return p;
}
po_problem rc_problem_handle(void *p) { return static_cast<RandomConvex *>(p)->handle(); }
void rc_problem_sizes(void *p, int64_t *nlocal, int64_t *offset) {
  *nlocal = static_cast<RandomConvex *>(p)->nlocal;
  *offset = static_cast<RandomConvex *>(p)->offset;
}
void rc_problem_set_linear_constraints(void *p, int flag) { static_cast<RandomConvex *>(p)->setLinearConstraints(flag); }
void rc_problem_set_deferred_reductions(void *p, int flag) { static_cast<RandomConvex *>(p)->setDeferredReductions(flag); }
double rc_problem_own_kernel_bytes(void *p, int jacobian_rewritten) {
  return static_cast<RandomConvex *>(p)->ownKernelBytes(jacobian_rewritten);
}
void rc_problem_destroy(void *p) { static_cast<RandomConvex *>(p)->decref(); }
}

#ifndef RANDOM_CONVEX_NO_MAIN
int main(int argc, char *argv[]) {
  long n = 1000000;
  int c = 32, iters = 30, deferred = 0;
  char qn[32] = "sr1";
  for (int k = 1; k < argc; k++) {
    sscanf(argv[k], "n=%ld", &n);
    sscanf(argv[k], "c=%d", &c);
    sscanf(argv[k], "iters=%d", &iters);
    sscanf(argv[k], "deferred=%d", &deferred);
    sscanf(argv[k], "qn=%31s", qn);
  }
  po_ctx ctx = NULL;
  if (po_ctx_create(0, &ctx) != 0) {
    fprintf(stderr, "no MI355X available: %s\n", po_last_error());
    return 2;
  }
  RandomConvex *prob = new RandomConvex(ctx, n, c, 0);
  prob->incref();
  prob->setDeferredReductions(deferred);
  ParOptOptions *options = new ParOptOptions();
  options->incref();
  // examples/random_convex/random_convex.py:116-126
  options->setOption("algorithm", "ip");
  options->setOption("qn_type", qn);
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
  std::vector<ParOptScalar> cons(c > 0 ? c : 1);
  ParOptScalar fobj = 0.0;
  prob->setDeferredReductions(0);
  prob->evalObjCon(x, &fobj, cons.data());
  printf("{\"rc\": %d, \"niter\": %d, \"neval\": %d, \"ngeval\": %d, \"fobj\": %.15e, \"xnorm\": %.15e, \"z0\": %.15e}\n",
         rc, niter, neval, ngeval, fobj, x->norm(), c > 0 ? z[0] : 0.0);
  opt->decref();
  options->decref();
  prob->decref();
  po_ctx_destroy(ctx);
  return rc;
}
#endif
