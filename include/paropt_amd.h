/*
 * paropt_amd -- C ABI of the MI355X-native interior-point hot path.
 *
 * This header is the drop-in boundary.  The reference (smdogroup/paropt v2.1.5) exposes this
 * path as a C++ virtual-class API (ParOptVec / ParOptCompactQuasiNewton / ParOptProblem /
 * ParOptInteriorPoint); its own FFI (the Cython layer) reaches it through C function-pointer
 * trampolines (src/CyParOptProblem.h:44-69).  Every entry point below names the reference
 * interface it replaces (file:line relative to the reference tree).  Signatures use plain
 * pointers, sizes and opaque handles only -- no C++ or torch types.
 *
 * Conventions
 *   - every function returns 0 on success and a non-zero PO_ERR_* code on failure; the text of
 *     the last failure on the calling thread is available from po_last_error().  Nothing aborts.
 *   - one po_ctx per process/GPU (one HIP stream, one communicator); all reducing calls are
 *     collective over the ranks of the context, like the reference's MPI calls.
 *   - vectors hold fp64 and live in HBM.  Ownership is the reference's intrusive refcount
 *     (src/ParOptVec.h:28-47): create returns a vector with count 1 (the reference's
 *     "create + incref" pair), po_vec_decref frees it at 0.
 */
#ifndef PAROPT_AMD_H
#define PAROPT_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct po_ctx_s *po_ctx;
typedef struct po_vec_s *po_vec;
typedef struct po_qn_s *po_qn;
typedef struct po_problem_s *po_problem;
typedef struct po_ip_s *po_ip;

enum {
  PO_OK = 0,
  PO_ERR_HIP = 1,       /* a HIP runtime call failed */
  PO_ERR_ARG = 2,       /* invalid argument / size mismatch */
  PO_ERR_COMM = 3,      /* RCCL / communicator failure */
  PO_ERR_NO_DEVICE = 4, /* no usable gfx950 device: the product has no CPU fallback */
  PO_ERR_OPTION = 5,    /* unknown option or wrong type (src/ParOptOptions.cpp:310-386) */
  PO_ERR_USER = 6,      /* a problem callback returned non-zero */
  PO_ERR_NUMERIC = 7    /* a factorization broke down (sparse Cholesky pivot <= 0) */
};

const char *po_last_error(void);
/* Library identification: "paropt_amd <version> gfx950". */
const char *po_version(void);

/* ---- context / communicator (replaces MPI_Comm of src/ParOptProblem.cpp:7-11) ------------- */
int po_ctx_create(int device, po_ctx *out);
int po_ctx_destroy(po_ctx ctx);
int po_ctx_synchronize(po_ctx ctx);
int po_ctx_rank(po_ctx ctx, int *rank, int *size);
/* The HIP stream every kernel of this context is launched on (a hipStream_t). */
void *po_ctx_stream(po_ctx ctx);
/* Diagnostics: host-synchronising reductions (= collectives when there is more than one rank) and kernel launches
 * issued on this context so far. */
int po_ctx_counters(po_ctx ctx, int64_t *reductions, int64_t *launches);
/* Diagnostics of the completion path of the reductions (round 5): the host does not synchronise the stream after a
 * reduction but polls a sequence number that the last kernel of the exchange -- the final reduction stage on one rank,
 * a one-workgroup publish kernel behind ncclAllReduce / ncclAllGather on several -- raises in pinned host memory behind
 * its results.  flag_waits = polled completions, flag_timeouts = those whose bounded spin (200 ms) ran out and fell
 * back to hipStreamSynchronize (always correct; a non-zero count in a soak run means the flag path is broken or a
 * single kernel ran longer than the bound), allreduces / allgathers = RCCL collectives issued. */
int po_ctx_sync_counters(po_ctx ctx, int64_t *flag_waits, int64_t *flag_timeouts, int64_t *allreduces,
                         int64_t *allgathers);
/* Diagnostics: algorithmic HBM bytes of every n-sized launch issued on this context so far -- each operand stream of
 * a launch counted once (8 n bytes), e.g. 8 (nvecs + 1) n for ParOptVec::mdot (src/ParOptVec.cpp:152-170), SURVEY.md
 * 8d's per-kernel figures.  `user` is the part issued from inside the problem's evaluation callbacks (the built-in
 * problems' kernels and any po_vec_* call a callback makes; a callback's own kernels are not seen).  bench.py's
 * iteration-level roofline is (bytes of the timed iterations) / time / HBM peak. */
int po_ctx_algorithmic_bytes(po_ctx ctx, double *total, double *user);
/* Leak check: device vectors currently alive in this process (every ParOptVec, the panels of the quasi-Newton
 * objects, the work vectors of the solvers) and the HBM bytes behind them.  The reference's intrusive
 * reference counts (src/ParOptVec.h:28-47) are kept, so after the last decref / destroy both return to their
 * values before the objects were made. */
int po_live_objects(int64_t *vectors, int64_t *bytes);
/* pinned host mirrors (po_vec_get_array) currently alive in this process */
int po_live_host_mirrors(int64_t *mirrors);
/* number of visible HIP devices (0 when there is none) */
int po_device_count(int *count);
/* Default option registries, for host-side ParOptOptions objects (src/ParOptOptions.h:9-61): the visitor is called
 * once per option of ParOptInteriorPoint::addDefaultOptions (which = 0, src/ParOptInteriorPoint.cpp:536-727),
 * ParOptTrustRegion::addDefaultOptions (1, src/ParOptTrustRegion.cpp:739-847) or ParOptMMA::addDefaultOptions
 * (2, src/ParOptMMA.cpp:234-289) with type 1 string / 2 boolean / 3 int / 4 float / 5 enum (the reference's
 * PAROPT_*_OPTION codes), the default and the admissible range / enum values. */
typedef void (*po_option_visitor)(void *user, const char *name, int type, const char *sval, int ival, int ilo,
                                  int ihi, double fval, double flo, double fhi, int nenum,
                                  const char *const *enumvals);
int po_options_visit_defaults(int which, po_option_visitor fn, void *user);
/* Live timing of the headline kernel inside a run: after po_ctx_time_mdot(ctx, nvecs) every ParOptVec::mdot
 * launch with exactly `nvecs` vectors on this context is bracketed by HIP events on the context's stream
 * (nvecs = 0 switches it off; every call resets the accumulators); the result call returns the accumulated
 * kernel milliseconds and the number of launches. */
int po_ctx_time_mdot(po_ctx ctx, int nvecs);
int po_ctx_time_mdot_result(po_ctx ctx, double *ms_total, int64_t *launches);
/* The same for the weighted-Gram launches of setUpKKTDiagSystem/setUpKKTSystem (src/ParOptInteriorPoint.cpp:
 * 1935-1950, 2648-2654): which = 0 the plain launches, 1 the launches that also form the L-SR1 columns;
 * ncols = panel width of the last timed launch, alg_bytes_total = algorithmic HBM bytes of the timed launches. */
int po_ctx_time_wgram(po_ctx ctx, int on);
int po_ctx_time_wgram_result(po_ctx ctx, int which, double *ms_total, int64_t *launches, int *ncols,
                             double *alg_bytes_total);
/* Communicator in use (0 self, 1 RCCL, 2 host callback) and the collectives issued so far: pure-sum
 * reductions go through ncclAllReduce, mixed SUM/MIN/MAX payloads through one rank-ordered ncclAllGather. */
int po_ctx_comm_info(po_ctx ctx, int *kind, int64_t *allreduces, int64_t *allgathers);
/* Reduction batching (default on; PAROPT_AMD_NO_BATCH=1 in the environment switches the default off): reductions
 * whose results are not needed one before the other -- the barrier sums at the line search's trial point with the
 * built-in problems' f and c (the MPI_Allreduce calls of src/ParOptInteriorPoint.cpp:3541-3565 and of the problem's
 * evalObjCon), the next residual's norms (:1588-1723) with the quasi-Newton products of update()
 * (src/ParOptQuasiNewton.cpp:162-186, 660-690) -- share ONE collective, one device-to-host copy and one host
 * synchronisation.  The values are the same bits either way; po_ctx_counters() counts host synchronisations,
 * po_ctx_batched_reductions() the reductions that rode along with another one.  Reductions issued by user code
 * through this ABI are never deferred. */
int po_ctx_set_reduction_batching(po_ctx ctx, int on);
int po_ctx_batched_reductions(po_ctx ctx, int64_t *batched);
/* Copy between a host buffer and a raw device array this library handed out (the Jacobian entries of
 * po_problem_set_sparse_jacobian_data), ordered with the context's stream; returns when the copy is done.
 * to_device != 0: host -> device. */
int po_ctx_memcpy(po_ctx ctx, void *dst, const void *src, int64_t bytes, int to_device);
/* RCCL over xGMI: rank 0 calls po_rccl_unique_id, ships the bytes to the other ranks by any
 * side channel (the Python harness uses the torch.distributed store), then every rank calls
 * po_ctx_comm_init_rccl.  Replaces the MPI_Allreduce/Reduce/Bcast sites of SURVEY.md 2.3. */
#define PO_RCCL_ID_BYTES 128
int po_rccl_unique_id(void *id128);
/* Also runs known-answer collectives over the solver's own exchange path before it returns (all-reduce of rank + 1
 * = N (N + 1) / 2; rank order of the all-gather; SUM / MIN / MAX combine) and checks the library's version against
 * the ABI its hand-declared prototypes assume; any mismatch is PO_ERR_COMM with the reason in po_last_error(). */
int po_ctx_comm_init_rccl(po_ctx ctx, int rank, int size, const void *id128);
/* version code of the loaded librccl (ncclGetVersion; 0 before the library was needed) */
int po_rccl_version(int *version_code);
/* MPI_Allreduce for user code on `count` host values, in place (op 0 SUM, 1 MIN, 2 MAX): what a problem's evalObjCon
 * does with its rank-local objective parts (examples/rosenbrock/rosenbrock.cpp:103-106).  Goes through the context's
 * communicator (RCCL or the host callback); a no-op on one rank.  Collective. */
int po_ctx_allreduce(po_ctx ctx, double *values, int count, int op);
/* Latency of one reduction exchange as the solver issues it (final-stage payload of `count` doubles -> collective
 * -> device-to-host copy -> host sync), `reps` times: out_us3 = {median, min, max} host microseconds.  pure_sum != 0:
 * the ncclAllReduce form, else the rank-ordered ncclAllGather form.  Collective. */
int po_ctx_bench_collective(po_ctx ctx, int count, int pure_sum, int reps, double *out_us3);
/* Host-side communicator hook: `allgather` must gather `count` doubles from every rank into
 * `out` (rank-major).  Lets a maintainer keep MPI (or gloo) underneath. */
typedef int (*po_allgather_fn)(const double *in, double *out, int count, void *user);
int po_ctx_comm_init_callback(po_ctx ctx, int rank, int size, po_allgather_fn fn, void *user);

/* ---- ParOptVec: src/ParOptVec.h:53-70, src/ParOptVec.cpp ---------------------------------- */
int po_vec_create(po_ctx ctx, int64_t nlocal, po_vec *out); /* ParOptBasicVec ctor :15-20 (zero-filled) */
int po_vec_incref(po_vec v);                                /* ParOptBase::incref  src/ParOptVec.h:34 */
int po_vec_decref(po_vec v);                                /* ParOptBase::decref  src/ParOptVec.h:38-43 */
int po_vec_size(po_vec v, int64_t *nlocal);
int po_vec_set(po_vec v, double alpha);                     /* set          :32-36 */
int po_vec_zero(po_vec v);                                  /* zeroEntries  :41-43 */
int po_vec_copy(po_vec dst, po_vec src);                    /* copyValues   :50-56 */
int po_vec_scale(po_vec v, double alpha);                   /* scale        :177-186 */
int po_vec_axpy(po_vec y, double alpha, po_vec x);          /* axpy         :191-204 */
int po_vec_dot(po_vec x, po_vec y, double *out);            /* dot          :124-143 */
int po_vec_mdot(po_vec x, const po_vec *vecs, int nvecs, double *out); /* mdot :152-170 (host out) */
int po_vec_norm(po_vec x, double *out);                     /* norm         :63-80 */
int po_vec_maxabs(po_vec x, double *out);                   /* maxabs       :87-99 */
int po_vec_l1norm(po_vec x, double *out);                   /* l1norm       :106-116 */
/* getArray :212-217 -- in the reference the pointer IS the data.  Here it is a pinned HOST mirror of the HBM
 * data that stays coherent through this ABI: po_vec_get_array downloads the vector (unless the mirror is already
 * live) and marks the mirror LIVE; while it is live every po_* call that reads the vector uploads the mirror
 * first and every po_* call that writes it downloads the result, so host writes are seen without an explicit
 * sync and the pointer stays valid for the life of the vector.  po_vec_release_array ends the live state (with
 * a final upload when `upload` != 0): REQUIRED for vectors the solver owns (callback arguments, optimized
 * points) before the solver runs on, because the solver's own kernels do not look at mirrors; optional for
 * vectors the caller created.  po_vec_sync_to_host is a plain download that does not make the mirror live;
 * po_vec_sync_to_device a plain upload.  The solver itself never uses host mirrors. */
int po_vec_get_array(po_vec v, double **host);
int po_vec_release_array(po_vec v, int upload);
/* the mirror after a fresh download, WITHOUT making it live (a read-only look, or a buffer for a write that is
 * followed by po_vec_sync_to_device) */
int po_vec_peek_array(po_vec v, double **host);
int po_vec_sync_to_device(po_vec v);
int po_vec_sync_to_host(po_vec v);
/* Device-resident problems use the raw HBM pointer instead. */
int po_vec_get_device_array(po_vec v, double **device);
/* y <- beta*y + sum_j alpha[j]*vecs[j]: the fused form of the reference's axpy loops
 * (src/ParOptQuasiNewton.cpp:414-416, src/ParOptInteriorPoint.cpp:1354-1356). */
int po_vec_maxpy(po_vec y, double beta, const double *alpha, const po_vec *vecs, int nvecs);
/* Deterministic counter-hash fill: v[i] = shift + scale*u01(seed, array_id, offset+i). */
int po_vec_fill_hash(po_vec v, uint64_t seed, uint64_t array_id, int64_t offset, double scale,
                     double shift);

/* ---- ParOptCompactQuasiNewton: src/ParOptQuasiNewton.h:32-220 ----------------------------- */
enum { PO_QN_BFGS = 0, PO_QN_SR1 = 1 };
enum { PO_BFGS_SKIP_NEGATIVE_CURVATURE = 0, PO_BFGS_DAMPED_UPDATE = 1 }; /* :10-13 */
enum { PO_QN_YTY_OVER_YTS = 0, PO_QN_YTS_OVER_STS = 1 };                 /* :18-23 */
int po_qn_create(po_ctx ctx, int type, int64_t nlocal, int subspace, po_qn *out); /* ctors .cpp:18-74, 496-556 */
int po_qn_destroy(po_qn qn);
int po_qn_set_update_type(po_qn qn, int bfgs_update_type);  /* setBFGSUpdateType .cpp:107-109 */
int po_qn_set_diag_type(po_qn qn, int diag_type);           /* setInitDiagonalType .cpp:116-119 */
int po_qn_reset(po_qn qn);                                  /* reset .cpp:127-142, 603-618 */
/* update(x,z,zw,s,y) .cpp:162-334, 636-747: returns the update type through *rc
 * (0 normal, 1 damped, 2 skipped).  s and y are not modified. */
int po_qn_update(po_qn qn, po_vec s, po_vec y, int *rc);
int po_qn_mult(po_qn qn, po_vec x, po_vec y);               /* mult .cpp:390-418, 760-778 */
int po_qn_mult_add(po_qn qn, double alpha, po_vec x, po_vec y); /* multAdd .cpp:432-459, 791-809 */
/* getCompactMat .cpp:471-487, 821-837: *size = k; b0; d0[k]; M[k*k] column-major; Z[k] borrowed
 * handles valid until the next update/reset.  Any output pointer may be NULL. */
int po_qn_get_compact(po_qn qn, int *size, double *b0, const double **d0, const double **M,
                      const po_vec **Z);
int po_qn_max_size(po_qn qn, int *size);                    /* getMaxLimitedMemorySize */
/* 0-based pivot rows of the LU factorization of the compact matrix M (the reference's LAPACK mfpiv,
 * src/ParOptQuasiNewton.cpp:375, 743, is 1-based); borrowed, valid until the next update / reset. */
int po_qn_get_pivots(po_qn qn, const int **mfpiv, int *n);
/* Test hook of the state-injected known-answer tests: loads msub pairs (device vectors of the local size, copied),
 * b0 and the small matrices the updates maintain -- B = S^T S, L = strictly lower triangle of S^T Y, D = diag(S^T Y),
 * column-major with leading dimension ld (src/ParOptQuasiNewton.h:141-147) -- and rebuilds (d0, M, LU, Z) from them
 * with the arithmetic of update() (computeMatUpdate :339-377; L-SR1 :712-743, columns left unformed). */
int po_qn_debug_load(po_qn qn, int msub, double b0, const double *B, const double *L, const double *D, int ld,
                     const po_vec *S, const po_vec *Y);

/* ---- ParOptProblem: src/ParOptProblem.h:42-296 -------------------------------------------- */
/* User problems are bound the way the reference's own FFI binds them: a table of C callbacks
 * (src/CyParOptProblem.h:44-69).  Callbacks receive device vectors; use
 * po_vec_get_device_array (device kernels) or po_vec_get_array + po_vec_sync_to_device (host). */
typedef struct {
  void *user;
  /* getVarsAndBounds :133 */
  int (*get_vars_and_bounds)(void *user, po_vec x, po_vec lb, po_vec ub);
  /* evalObjCon :146-158 -- fobj and cons[ncon] must be identical on all ranks */
  int (*eval_obj_con)(void *user, po_vec x, double *fobj, double *cons);
  /* evalObjConGradient :171-172 */
  int (*eval_obj_con_gradient)(void *user, po_vec x, po_vec g, const po_vec *Ac);
  /* computeQuasiNewtonUpdateCorrection :211-213 (may be NULL) */
  int (*qn_update_correction)(void *user, po_vec x, const double *z, po_vec s, po_vec y);
  /* writeOutput :289 (may be NULL) */
  int (*write_output)(void *user, int iter, po_vec x);
} po_problem_callbacks;
int po_problem_create_callbacks(po_ctx ctx, int64_t nlocal, int ncon, int ninequality,
                                const po_problem_callbacks *cb, po_problem *out);
/* Sparse ("weighting") constraints with block-diagonal Aw D^-1 Aw^T and nwblock = 1
 * (src/ParOptProblem.h:215-262, src/ParOptSparseMat.cpp:11-229).  `out`, `pzw` and `A` are
 * w-sized device vectors (nwcon local entries); A is the diagonal the reference passes as a raw
 * array.  The first nwinequality local constraints are inequalities (cw >= 0). */
typedef struct po_problem_sparse_callbacks {
  /* evalSparseCon :215: out = cw(x) */
  int (*eval_sparse_con)(void *user, po_vec x, po_vec out);
  /* addSparseJacobian :226: out += alpha * Aw(x) * px */
  int (*add_sparse_jacobian)(void *user, double alpha, po_vec x, po_vec px, po_vec out);
  /* addSparseJacobianTranspose :239: out += alpha * Aw(x)^T * pzw */
  int (*add_sparse_jacobian_transpose)(void *user, double alpha, po_vec x, po_vec pzw, po_vec out);
  /* addSparseInnerProduct :252: A += alpha * diag(Aw(x) * diag(cvec) * Aw(x)^T) */
  int (*add_sparse_inner_product)(void *user, double alpha, po_vec x, po_vec cvec, po_vec A);
} po_problem_sparse_callbacks;
/* setProblemSizes / setNumInequalities for the sparse block (src/ParOptProblem.h:88-96); only
 * valid on a problem made by po_problem_create_callbacks, before po_ip_create */
int po_problem_set_sparse_callbacks(po_problem p, int64_t nwcon, int64_t nwinequality,
                                    const po_problem_sparse_callbacks *cb);
/* CSR form of the sparse constraints: ParOptSparseProblem (src/ParOptProblem.h:301-395,
 * src/ParOptProblem.cpp:624-816; Cython side src/CyParOptProblem.h:177-262).  The Jacobian Aw of the
 * nwcon rank-local sparse constraints has a FIXED pattern (rowp[nwcon+1], cols[nnz], local column
 * indices, any order within a row, no duplicates); rows may overlap, so S = C + Aw D^-1 Aw^T is a general
 * sparse SPD matrix, factored on the device (ParOptQuasiDefSparseMat, src/ParOptSparseMat.cpp:234-450).
 *   eval_sparse_obj_con          evalSparseObjCon :332: also writes sparse_con = cw(x) (w-sized device vector)
 *   eval_sparse_obj_con_gradient evalSparseObjConGradient :335: also writes the nnz Jacobian entries, in the
 *                                order of `cols`, to `data` - a DEVICE array (hipMemcpy or a kernel)
 * Replaces the problem's eval_obj_con / eval_obj_con_gradient callbacks, as the reference's subclass does.
 * Only on a problem made by po_problem_create_callbacks, before po_ip_create. */
typedef int (*po_eval_sparse_obj_con_fn)(void *user, po_vec x, double *fobj, double *cons, po_vec sparse_con);
typedef int (*po_eval_sparse_obj_con_gradient_fn)(void *user, po_vec x, po_vec g, const po_vec *Ac,
                                                  double *data, int64_t nnz);
int po_problem_set_sparse_jacobian_data(po_problem p, int64_t nwcon, int64_t nwinequality, const int *rowp,
                                        const int *cols, po_eval_sparse_obj_con_fn eval_sparse_obj_con,
                                        po_eval_sparse_obj_con_gradient_fn eval_sparse_obj_con_gradient);
/* getSparseJacobianData (src/ParOptProblem.cpp:689-703): host pattern, device values; returns nnz in *nnz.
 * Borrowed pointers, valid until the next po_problem_set_sparse_jacobian_data or the problem's destruction -- also
 * across the library's own change of path when a recognised grouped pattern turns out to have non-uniform entries. */
int po_problem_get_sparse_jacobian_data(po_problem p, const int **rowp, const int **cols, double **data,
                                        int64_t *nnz);
/* ParOptQuasiDefMat (src/ParOptSparseMat.h:18-62) of any problem with sparse constraints, block or CSR form:
 * factor: c holds the diagonal C on entry (the block form leaves 1/(C + diag(Aw dinv Aw^T)) in it); returns
 * PO_ERR_NUMERIC when the sparse Cholesky met a non-positive pivot (D^-1 or C not positive) - inside the
 * interior point the same event is counted, warned about once and survived, as the reference ignores
 * LAPACK's info there (src/ParOptSparseCholesky.cpp:631);
 * apply: [D Aw^T; Aw -C] [yx; -yw] = [bx; bw] with the factor of the last po_quasidef_factor, bw may be
 * NULL (the three-argument apply :39).  bx must not alias yx. */
int po_quasidef_factor(po_problem p, po_vec x, po_vec dinv, po_vec c);
int po_quasidef_apply(po_problem p, po_vec x, po_vec dinv, po_vec c, po_vec bx, po_vec bw, po_vec yx, po_vec yw);
/* getFactorInfo (src/ParOptSparseMat.cpp:433-450): one line about the sparse factor, NULL for the block form */
const char *po_quasidef_factor_info(po_problem p);
/* The one-time host analysis behind po_problem_set_sparse_jacobian_data, exposed so it can be inspected (and
 * tested) without a device: column-sorted pattern, pattern of S = Aw Aw^T, nested-dissection ordering,
 * elimination tree, pattern of L (CSR, diagonal last in each row) and the dependency level sets the device
 * factorization and solves are scheduled by.  No context needed.
 * info = {nnz(Aw), nnz(lower S), nnz(L), dependency levels, 1 if `cols` was already sorted, number of fronts,
 * rows of the largest front}. */
typedef struct po_csr_symbolic_s *po_csr_symbolic;
int po_csr_symbolic_create(int64_t nvars, int64_t nwcon, const int *rowp, const int *cols, po_csr_symbolic *out);
int po_csr_symbolic_info(po_csr_symbolic h, int64_t info[7]);
/* borrowed host arrays: perm[new] = old (nwcon), parent (nwcon), Lrowp (nwcon+1), Lcols (nnz(L)), level_ptr
 * (levels+1), front_of (nwcon).  Rows are numbered level by level; level l is the rows level_ptr[l] ..
 * level_ptr[l+1]-1, processed in ascending order by the factorization and the forward solve and in descending
 * order by the backward solve.  Inside a level come first the ordinary rows (front_of = -1), which depend on
 * earlier levels only, then its FRONTS: dense separator cliques whose rows are contiguous, front_of = first row
 * of the front, row f0 + r ending with the columns f0 .. f0 + r (the part left of f0 depends on earlier levels
 * only; the triangle inside is factored / solved by one workgroup).  Any output pointer may be NULL. */
int po_csr_symbolic_arrays(po_csr_symbolic h, const int **perm, const int **parent, const int **Lrowp,
                           const int **Lcols, const int **level_ptr, const int **front_of);
int po_csr_symbolic_destroy(po_csr_symbolic h);
/* nwblock of ParOptQuasiDefBlockMat (src/ParOptSparseMat.cpp:11-229) for the callback form: consecutive blocks of
 * nwblock (1..16) sparse constraints may share variables inside a block.  add_sparse_inner_product then receives,
 * instead of the w-sized diagonal, the packed upper triangles of the nwblock x nwblock blocks - entry (i, j),
 * i <= j, of block b at b nwblock (nwblock+1)/2 + i + j (j+1)/2, nwcon (nwblock+1)/2 entries in all - as the
 * reference passes them (src/ParOptProblem.h:252-262).  After po_problem_set_sparse_callbacks. */
int po_problem_set_sparse_block_size(po_problem p, int nwblock);
/* Second-order information (src/ParOptProblem.h:160-189) for use_hvec_product / use_diag_hessian:
 * evalHvecProduct: hvec = H(x, z, zw) px ; evalHessianDiag: hdiag = diag H(x, z, zw), with H the
 * Hessian of the Lagrangian f - z^T c - zw^T cw.  zw is NULL when nwcon = 0.  Either may be NULL. */
typedef int (*po_hvec_fn)(void *user, po_vec x, const double *z, po_vec zw, po_vec px, po_vec hvec);
typedef int (*po_hdiag_fn)(void *user, po_vec x, const double *z, po_vec zw, po_vec hdiag);
int po_problem_set_hessian_callbacks(po_problem p, po_hvec_fn hvec, po_hdiag_fn hdiag);
/* Built-in device-resident workloads of BASELINE.json (DESIGN.md "Workloads"): */
enum { PO_PROBLEM_QUADRATIC = 0, PO_PROBLEM_CONVEX = 1, PO_PROBLEM_ROSENBROCK = 2 };
int po_problem_create_separable(po_ctx ctx, int kind, int64_t nglobal, int ncon, uint64_t seed,
                                double eig_min, double eig_max, po_problem *out);
/* Weighting constraints on a built-in workload (BASELINE.json configs[3]; the pattern of
 * examples/rosenbrock/rosenbrock.cpp:131-184): cw_i = 1 - sum_{k<nw} x[nwstart + i (nw+nwskip) + k],
 * i < nwcon (global indices), the first nwinequality of them inequalities.  Before po_ip_create. */
int po_problem_set_weighting(po_problem p, int64_t nwcon, int nw, int64_t nwstart, int nwskip,
                             int64_t nwinequality);
int po_problem_sparse_sizes(po_problem p, int64_t *nwcon_local, int64_t *nwinequality_local);
/* Overlapping nonlinear sparse constraints on a built-in workload, in the CSR form above and rank-local like
 * examples/rosenbrock/sparse_rosenbrock.cpp:38-118 (which is span 2, stride 1):
 * cw_i = 1 - sum_{k<span} x[i*stride + k]^2 >= 0, i < (nlocal - span)/stride + 1.  reverse_cols stores each
 * row's columns in descending order (exercises the unsorted-pattern path).  Before po_ip_create. */
int po_problem_set_chain(po_problem p, int span, int stride, int reverse_cols);
/* useLowerBounds / useUpperBounds (src/ParOptProblem.h:140-150; CyParOptProblem::setVarBoundOptions):
 * a problem that declares a side unused never has that side's bound multipliers formed.  Before
 * po_ip_create. */
int po_problem_set_var_bound_options(po_problem p, int use_lower, int use_upper);
/* Opt-in extension (not in the reference): DEFERRED REDUCTIONS for a callback problem.  By default every reduction
 * a callback issues through this ABI returns its value immediately (one collective + host synchronisation each, the
 * reference's MPI_Allreduce semantics).  With flag != 0 the problem promises that its evaluation callbacks
 *   - obtain reduced values ONLY through po_vec_dot / po_vec_mdot / po_vec_norm / ... and po_ctx_reduce_device,
 *     with result pointers that stay valid until the solver consumes them (the `fobj` / `cons` arguments of
 *     eval_obj_con are such pointers),
 *   - do not READ those results before returning, and put any host post-processing of them (e.g.
 *     cons[j] = beta[j] - cons[j]) into a hook registered with po_ctx_after_reduce,
 * so that the solver may queue them with its own reductions of the same step (the barrier sums at a line-search
 * trial point; the next residual's norms) and pay ONE collective + host synchronisation for all of them.  The
 * values are the same bits either way.  Outside the solver's batches the calls stay immediate. */
int po_problem_set_deferred_reductions(po_problem p, int flag);
/* Host work that depends on the results of reductions queued so far on this context: runs right away when nothing
 * is queued, else when the queue is flushed (in registration order). */
typedef void (*po_after_reduce_fn)(void *user);
int po_ctx_after_reduce(po_ctx ctx, po_after_reduce_fn fn, void *user);
/* Reduce `count` rank-local values that live in DEVICE memory (e.g. the last stage of a problem's own reduction
 * kernel) across the ranks of the context: op 0 SUM, 1 MIN, 2 MAX; host_out[count] receives the result --
 * immediately, or at the flush when the problem has deferred reductions and the solver has a batch open.  The
 * device-side counterpart of MPI_Allreduce; ordered with the context's stream.  Collective. */
int po_ctx_reduce_device(po_ctx ctx, const double *device_values, int count, int op, double *host_out);
/* Declares that the DENSE constraints are linear in x (their Jacobian is constant).  The reference's contract
 * (src/ParOptProblem.h:146-158) has evalObjConGradient rewrite all ncon gradient vectors at every call; with this
 * flag the solver keeps the Jacobian of the first evaluation of each optimize() call and afterwards invokes the
 * gradient callback with Ac == NULL ("objective gradient only").  Off by default: without it Ac is never NULL.
 * Rejected (PO_ERR_ARG) for the built-in Rosenbrock problem, whose constraints are not linear. */
int po_problem_set_linear_constraints(po_problem p, int flag);
/* Test data for initAndCheckDesignAndBounds on the built-in problems (oracle/ref_driver.cpp SepProblem::bounds_mode),
 * by global index gi: bit 1: gi % 7 == 3 -> lb = ub = midpoint; bit 2: gi % 11 == 5 -> x = lb; bit 4: gi % 13 == 6 ->
 * x = ub. */
int po_problem_set_bounds_mode(po_problem p, int mode);
int po_problem_destroy(po_problem p);
int po_problem_sizes(po_problem p, int64_t *nlocal, int64_t *offset, int *ncon);
int po_problem_eval_obj_con(po_problem p, po_vec x, double *fobj, double *cons);
int po_problem_eval_obj_con_gradient(po_problem p, po_vec x, po_vec g, const po_vec *Ac);
int po_problem_get_vars_and_bounds(po_problem p, po_vec x, po_vec lb, po_vec ub);

/* ---- ParOptInteriorPoint: src/ParOptInteriorPoint.h:128-217 ------------------------------- */
int po_ip_create(po_problem prob, po_ip *out);              /* ctor .cpp:182-450 (default options) */
int po_ip_destroy(po_ip ip);
/* ParOptOptions::setOption overloads, src/ParOptOptions.h:35-37; names/defaults of
 * ParOptInteriorPoint::addDefaultOptions .cpp:536-727.  Must be called before po_ip_optimize;
 * qn_type / qn_subspace_size must be set before the first optimize. */
int po_ip_set_option_str(po_ip ip, const char *name, const char *value);
int po_ip_set_option_int(po_ip ip, const char *name, int value);
int po_ip_set_option_float(po_ip ip, const char *name, double value);
int po_ip_optimize(po_ip ip, const char *checkpoint);       /* optimize .cpp:4399-5333 */
/* getOptimizedPoint .cpp:793-826: borrowed handles (NULL allowed) */
int po_ip_get_optimized_point(po_ip ip, po_vec *x, const double **z, po_vec *zl, po_vec *zu);
/* getOptimizedSlacks .cpp:848-866 (+ the slack multipliers) */
int po_ip_get_optimized_slacks(po_ip ip, const double **s, const double **t, const double **zs,
                               const double **zt);
/* the w-sized blocks zw, sw, tw, zsw, ztw of the optimized point (NULL handles when nwcon = 0) */
int po_ip_get_optimized_sparse(po_ip ip, po_vec *zw, po_vec *sw, po_vec *tw, po_vec *zsw,
                               po_vec *ztw);
int po_ip_get_counters(po_ip ip, int *niter, int *neval, int *ngeval); /* getIterationCounters .h:203-217 */
int po_ip_get_barrier_parameter(po_ip ip, double *mu);      /* .cpp:1110 */
int po_ip_get_complementarity(po_ip ip, double *comp);      /* .cpp:1118-1120 */
int po_ip_get_objective(po_ip ip, double *fobj, double *rho);
int po_ip_set_penalty_gamma(po_ip ip, double gamma);        /* .cpp:1127-1151 */
/* setPenaltyGamma(const double*) .cpp:1160-1172: one value per dense constraint (negative = keep) */
int po_ip_set_penalty_gamma_array(po_ip ip, const double *gamma);
/* setQuasiNewton .cpp:1193-1234: use a caller-owned approximation (NULL detaches it; the solver then
 * needs sequential_linear_method or use_diag_hessian); resetProblemInstance .cpp:745-764: swap in a
 * problem of identical sizes (the trust-region and MMA drivers do both) */
int po_ip_set_quasi_newton(po_ip ip, po_qn qn);
int po_ip_reset_problem_instance(po_ip ip, po_problem prob);
/* number of Hessian-vector products of the last optimize (getIterationCounters' 4th output) */
int po_ip_get_hvec_count(po_ip ip, int *nhvec);
int po_ip_reset_design_and_bounds(po_ip ip);                /* .cpp:1249-1251 */
/* checkGradients(dh) (.h:166, .cpp:6196-6199): the problem's finite-difference check at the solver's current point;
 * *report (borrowed, valid until the next call) holds the text the reference prints */
int po_ip_check_gradients(po_ip ip, double dh, const char **report);
/* checkMeritFuncGradient(xpt, dh) (.h:199, .cpp:3280-3432): prints the reference's two lines on rank 0;
 * xpt may be NULL (the current point and step are used); fd / actual may be NULL */
int po_ip_check_merit_func_gradient(po_ip ip, po_vec xpt, double dh, double *fd, double *actual);
int po_ip_reset_quasi_newton(po_ip ip);                     /* resetQuasiNewtonHessian .cpp:1241-1245 */
int po_ip_get_quasi_newton(po_ip ip, po_qn *qn);            /* borrowed */
/* collective: ONE file in the reference's MPI-IO layout whatever the rank count (rank 0 writes the header and
 * the dense blocks, every rank its block of x, zl, zu, zw, sw at its global offset) */
int po_ip_write_solution_file(po_ip ip, const char *filename); /* .cpp:883-972 */
int po_ip_read_solution_file(po_ip ip, const char *filename);  /* .cpp:983-1104 (restart) */
/* Per-iteration observer, called at the point the reference calls prob->writeOutput
 * (.cpp:4620-4630); used by the parity tests to snapshot the state.  An OBSERVER: it may read the iterate (also through
 * po_vec_get_array views) but must not write it mid-solve -- sums of the current point are carried between passes. */
typedef int (*po_ip_iteration_fn)(void *user, int iter);
int po_ip_set_iteration_callback(po_ip ip, po_ip_iteration_fn fn, void *user);
/* The iteration table of the last optimize() in the reference's paropt.out column layout
 * (.cpp:4777-4801); *text is owned by the solver. */
int po_ip_get_history(po_ip ip, const char **text);
/* Time (seconds, HIP events on the context stream) spent per phase during the last optimize;
 * names is a ';'-separated list matching seconds[]. */
int po_ip_get_phase_times(po_ip ip, const char **names, const double **seconds, int *count);
/* The "user_eval" entry of po_ip_get_phase_times -- stream time of the problem's callbacks -- needs a pair of event
 * records around every callback; each is a packet between two kernels and costs the stream some dispatch latency
 * (1.5 % of an inner iteration at n = 5 M, nothing measurable at n >= 10 M), so it is OFF by default: on != 0 switches
 * it on for the solves that follow (bench.py does, for the figure it reports).  No reference counterpart. */
int po_ip_set_callback_timing(po_ip ip, int on);
/* Single-step entry points used by the known-answer tests (reference private methods
 * computeKKTRes/setUpKKTDiagSystem/setUpKKTSystem/computeKKTStep, .cpp:1337, 1832, 2634, 2700):
 * computes the KKT step at the current state with barrier mu into internal step storage and
 * returns borrowed handles / pointers to it. */
/* Integer bookkeeping of the iteration (SURVEY 8a'), for bit-exact comparison with the reference:
 *   gpiv[ngpiv]   0-based pivot rows of the LU factorization of the dense Schur complement G of the last
 *                 setUpKKTDiagSystem (the reference's LAPACK gpiv, src/ParOptInteriorPoint.cpp:1968-1969, is 1-based;
 *                 cpiv is not comparable: the product factors Ce from one Gram matrix);
 *   check_flag    OR of the bound-repair bits of initAndCheckDesignAndBounds (:4290-4344): 1 inconsistent bounds,
 *                 2 / 4 variables moved away from the lower / upper bound;
 *   clamped[8]    entries sitting exactly at their clamp values (:3150-3190, 4177-4195): x == lb + eps,
 *                 x == ub - eps, zl == eps, zu == eps (global counts), then s, t, zs, zt == eps. Collective. */
int po_ip_get_debug_ints(po_ip ip, const int **gpiv, int *ngpiv, int *check_flag, int64_t clamped[8]);
/* borrowed lb / ub vectors as repaired by initAndCheckDesignAndBounds */
int po_ip_get_bounds(po_ip ip, po_vec *lb, po_vec *ub);
int po_ip_debug_kkt_step(po_ip ip, double mu, po_vec *px, po_vec *pzl, po_vec *pzu,
                         const double **pz, const double **ps, const double **pt,
                         const double **pzs, const double **pzt);
/* State-injected known-answer tests (tests/test_gpu_kat.py): the REFERENCE's state at one iteration -- the dump of
 * its private members made by oracle/ref_driver.cpp -- is loaded into the device solver and the pieces of the KKT
 * step are compared one by one with what the reference's private methods produced from that same state
 * (computeKKTRes .cpp:1337, setUpKKTDiagSystem :1832-1971, setUpKKTSystem :2634-2667, computeKKTStep :2700-2737,
 * the refinement loop :4985-4991).  Usage, after one optimize() call has initialised the solver: write x, zl, zu (and
 * the sparse blocks) through the borrowed handles of po_ip_get_optimized_point / _sparse, load the limited-memory
 * pairs with po_qn_debug_load on the handle of po_ip_get_quasi_newton, then po_ip_debug_set_state (dense blocks and
 * the barrier parameter; evaluates the problem at x) and po_ip_debug_kkt.
 *   mode 0: ONE bordered solve with the quasi-Newton correction (computeKKTStep), the stored-step kernels;
 *   mode 1: the kernel sequence of a plain quasi-Newton iteration of optimize() (dinv_d1, fused
 *           Gram pass over unformed L-SR1 columns, first solve pass, refinement pass) = the step after ONE refinement.
 *   mode 2: the predictor-corrector step of optimize() (.cpp:4956-5045; no sparse constraints): affine solve with one
 *           refinement, probe to the boundary, the Mehrotra rule, corrector right-hand side (:1729-1789) and solve.
 *           The new barrier parameter is po_ip_get_barrier_parameter afterwards; step_mins are the corrector step's
 *           for the fraction to the boundary of that parameter (max(min_fraction_to_boundary, 1 - mu)).
 * Everything in the dump is borrowed and valid until the next call on the solver.  G and Ce are the Schur complements
 * AS ASSEMBLED (column-major c x c and k x k, before their LU factorizations), W the weighted Gram matrix
 * [Ac | Z]^T Dinv [Ac | Z] ((c+k) x (c+k)); gpiv / cpiv 0-based LU pivot rows; res_norms = max_prime, max_dual,
 * max_infeas, res_norm (computeResNorm :1588-1723); step_mins = the vector part of computeMaxStep (:2942-3103). */
typedef struct po_ip_kkt_dump {
  int c, k;
  po_vec Dinv, res_x;
  const double *res_z, *res_s, *res_t, *res_zs, *res_zt;
  double res_norms[4];
  const double *W, *G, *Ce;
  const int *gpiv, *cpiv;
  po_vec px, pzl, pzu;
  const double *pz, *ps, *pt, *pzs, *pzt;
  double step_mins[2];
} po_ip_kkt_dump;
int po_ip_debug_set_state(po_ip ip, const double *z, const double *s, const double *t, const double *zs,
                          const double *zt, double mu);
int po_ip_debug_kkt(po_ip ip, double mu, int mode, double tau, po_ip_kkt_dump *out);
/* the sparse blocks of that step (borrowed; NULL handles when the problem has no sparse constraints) */
int po_ip_debug_kkt_step_sparse(po_ip ip, po_vec *pzw, po_vec *psw, po_vec *ptw, po_vec *pzsw, po_vec *pztw);

/* ---- standalone hot kernels for the roofline bench ----------------------------------------- */
/* W = P^T diag(d) P, P = [vecs], column-major nvecs x nvecs on the host (MFMA fp64). */
int po_wgram(po_vec d, const po_vec *vecs, int nvecs, double *W);
/* The same pass with the LAST vector t pre-weighted: W[i][nvecs-1] = W[nvecs-1][i] = vecs[i] . t for i < nvecs-1
 * (the panel dots P^T t of the bordered solve that follows setUpKKTSystem ride in the Gram pass,
 * src/ParOptInteriorPoint.cpp:2139-2147); W[nvecs-1][nvecs-1] = t . t. */
int po_wgram_with_rhs(po_vec d, const po_vec *vecs, int nvecs, double *W);
/* Structured sparse Jacobian (one constraint per group of `nw` consecutive variables, period nw + skip, first group at
 * variable 0 -- the pattern of examples/rosenbrock/rosenbrock.cpp:131-184): the panel image U_j = alpha * (group sums
 * of d o vecs_j), j < ncols, which the reference forms column by column through ParOptProblem::addSparseJacobian
 * (src/ParOptProblem.h:215-262).  po_group_panel is the pass of its own; po_wgram_with_groups lets it ride in the Gram
 * pass over the same panel (W as po_wgram / po_wgram_with_rhs) and reports in *fused whether the fused kernel covered
 * this shape (0: U is left untouched).  Same bits from both. */
int po_group_panel(po_vec d, const po_vec *vecs, int ncols, int64_t nwcon, int nw, int skip, double alpha,
                   const po_vec *U);
int po_wgram_with_groups(po_vec d, const po_vec *vecs, int nvecs, int preweighted_last, int64_t nwcon, int nw,
                         int skip, double alpha, const po_vec *U, int ncols, double *W, int *fused);
/* Launch mdot `reps` times back to back on the context stream and return the average kernel
 * time in milliseconds measured with HIP events on that stream (bench.py's roofline leg). */
int po_bench_mdot(po_vec x, const po_vec *vecs, int nvecs, int reps, double *avg_ms, double *out);
int po_bench_wgram(po_vec d, const po_vec *vecs, int nvecs, int reps, double *avg_ms);
/* Same-run stream ceilings on the context stream: kind 0 = read-only (x.y, 16 B per element),
 * kind 1 = copy y <- x (8 B read + 8 B written per element); average kernel milliseconds over `reps`. */
int po_bench_stream(po_vec x, po_vec y, int kind, int reps, double *avg_ms);
/* Every hot kernel of one interior-point iteration (n local variables, c dense constraints, k <= 12
 * quasi-Newton columns) timed in isolation on synthetic vectors; `report` receives a JSON array of
 * {kernel, avg_ms, min_ms, alg_GB, GBps, frac_hbm_8TBps, TFLOPs} (tools/microbench.py). */
int po_bench_kernels(po_ctx ctx, int64_t n, int c, int k, int reps, char *report, int report_len);
/* Roofline table of the vector API and the quasi-Newton products at size n (tools/microbench.py --vec-api): every
 * ParOptVec operation (src/ParOptVec.cpp:32-204), mdot / the multi-vector axpy at 10 and 40 vectors, LBFGS::mult /
 * multAdd with 20 pairs and LSR1::mult / multAdd with 10 (src/ParOptQuasiNewton.cpp:390-459, 760-809), each with
 * SURVEY.md 8d's algorithmic bytes, the HIP-event time of `reps` back-to-back calls, and beside it the measured
 * ceiling of its stream mix (a trivial kernel moving the same input / output streams).  JSON array in `report`. */
int po_bench_vec_api(po_ctx ctx, int64_t n, int reps, char *report, int report_len);

/* ---- ParOptTrustRegion over the quadratic / compact-eigenvalue subproblem ------------------------
 * src/ParOptTrustRegion.h:376-480, set up as ParOptOptimizer does for algorithm = "tr"
 * (src/ParOptOptimizer.cpp:108-183): quasi-Newton object from qn_type / qn_subspace_size /
 * qn_update_type / qn_diag_type, ParOptQuadraticSubproblem (or ParOptEigenSubproblem), an
 * interior-point solver on the subproblem and the SL1QP driver.  ONE options registry holds the
 * interior-point options (.cpp:536-727) and the trust-region options
 * (src/ParOptTrustRegion.cpp:739-847), as in the reference.  `prob` is borrowed. */
typedef struct po_tr_s *po_tr;
typedef struct po_eig_s *po_eig;
int po_tr_create(po_problem prob, po_tr *out);
int po_tr_destroy(po_tr tr);
int po_tr_set_option_str(po_tr tr, const char *name, const char *value);
int po_tr_set_option_int(po_tr tr, const char *name, int value);
int po_tr_set_option_float(po_tr tr, const char *name, double value);
/* ParOptEigenSubproblem::setEigenModelUpdate (src/ParOptCompactEigenvalueApprox.h:166-170):
 * constraint `index` is modelled as c0 + g0^T s + 1/2 s^T H M H^T s with N directions; `update` is
 * called at the initial point and at every accepted point with c0, g0 preset to the linearisation and
 * must fill hvecs, M and Minv (row-major N x N).  Call before the first po_tr_optimize. */
typedef int (*po_eig_update_fn)(void *user, po_vec x, po_eig approx);
int po_tr_set_eigen_model(po_tr tr, int N, int index, po_eig_update_fn update, void *user);
/* built-in synthetic model of BASELINE.json configs[4]: unit hash directions (seed, array ids 300+i),
 * M = -curv (1 + 0.1 i) I (oracle/ref_driver.cpp eig_update) */
int po_tr_set_eigen_model_synthetic(po_tr tr, int N, int index, uint64_t seed, double curv);
/* ParOptCompactEigenApprox::getApproximation (.cpp:66-90): borrowed pointers */
int po_eig_get_approximation(po_eig approx, double **c0, po_vec *g0, int *N, double **M, double **Minv,
                             const po_vec **hvecs);
int po_tr_optimize(po_tr tr);                                /* optimize .cpp:2365-2384 */
/* getOptimizedPoint .cpp:872-876 (+ the multipliers of the last subproblem solve) */
int po_tr_get_optimized_point(po_tr tr, po_vec *x, const double **z, po_vec *zw);
/* driver state: radius, iteration count, subproblem iteration counts of the last iteration,
 * penalty parameters (borrowed), model values fk / ck (borrowed) */
int po_tr_get_state(po_tr tr, double *tr_size, int *iter_count, int *subproblem_iters,
                    int *adaptive_subproblem_iters, const double **penalty_gamma, double *fk,
                    const double **ck);
/* the last row of the iteration table (12 numeric columns without the wall time) and its info string */
int po_tr_get_last_row(po_tr tr, const double **row12, const char **info);
/* How the two interior-point solves of the latest trust-region iteration ENDED: the last line of each solve's
 * iteration table (src/ParOptInteriorPoint.cpp:4777-4801; steering / restoration solve, then the QP; empty when the
 * solve did not run).  Borrowed, valid until the next iteration. */
int po_tr_get_last_solve_lines(po_tr tr, const char **steering, const char **qp);
int po_tr_get_history(po_tr tr, const char **text);          /* the paropt.tr table :1406-1438 */
int po_tr_get_quasi_newton(po_tr tr, po_qn *qn);             /* subproblem->getQuasiNewton() */
int po_tr_get_model_vectors(po_tr tr, po_vec *xk, po_vec *gk);
typedef int (*po_tr_iteration_fn)(void *user, int iter);     /* where the reference calls writeOutput */
int po_tr_set_iteration_callback(po_tr tr, po_tr_iteration_fn fn, void *user);

/* ---- the same layer piece by piece, as the reference's user code assembles it -------------------------------
 * (examples/eigenvalue/eigenvalue_opt.py:298-308, src/ParOptOptimizer.cpp:108-183, 226-237):
 *   qn = ParOptLBFGS(problem, m)                         po_qn_create
 *   approx = ParOptCompactEigenApprox(problem, N)        po_eig_create
 *   eig_qn = ParOptEigenQuasiNewton(qn, approx, index)   po_eigqn_create      (a po_qn like any other)
 *   sub = ParOptEigenSubproblem(problem, eig_qn)         po_trsub_create_eigen (or _quadratic(problem, qn))
 *   sub->setEigenModelUpdate(data, fn)                   po_trsub_set_eigen_model_update
 *   ip = ParOptInteriorPoint(sub, options)               po_trsub_problem + po_ip_create
 *   tr = ParOptTrustRegion(sub, options)                 po_tr_create_subproblem
 *   tr->optimize(ip)                                     po_tr_optimize_with
 * Every argument is borrowed: the caller keeps the objects alive for as long as the objects built on them live,
 * and destroys them in reverse order (the reference's incref/decref chain, src/ParOptTrustRegion.cpp:660-737). */
/* ParOptCompactEigenApprox (src/ParOptCompactEigenvalueApprox.h:7-32, .cpp:23-120) */
int po_eig_create(po_problem prob, int N, po_eig *out);
int po_eig_destroy(po_eig approx);
int po_eig_mult_add(po_eig approx, double alpha, po_vec x, po_vec y);            /* y += alpha H M H^T x  .cpp:52-64 */
int po_eig_eval_approximation(po_eig approx, po_vec s, po_vec t, double *value); /* .cpp:92-106 (s or t NULL: c0) */
int po_eig_eval_approximation_gradient(po_eig approx, po_vec s, po_vec grad);    /* .cpp:108-120 */
/* ParOptEigenQuasiNewton(qn, eigh, index) (.h:34-84, .cpp:122-291): B = B_qn - z0 H M H^T as ONE compact matrix over
 * [Z_qn | H]; `qn` may be NULL.  The result is a po_qn: po_qn_mult / mult_add / get_compact / max_size / reset /
 * destroy apply; po_qn_update is the no-op of .cpp:176-179. */
int po_eigqn_create(po_qn qn, po_eig approx, int index, po_qn *out);
int po_eigqn_set_use_quasi_newton_objective(po_qn eig_qn, int truth);            /* .cpp:164-166 */
int po_eigqn_update_multipliers(po_qn eig_qn, const double *z);                  /* update(x, z, zw) .cpp:181-187 */
int po_eigqn_get_multiplier_index(po_qn eig_qn, int *index);
/* ParOptTrustRegionSubproblem (src/ParOptTrustRegion.h:15-151) in its two library forms */
typedef struct po_trsub_s *po_trsub;
int po_trsub_create_quadratic(po_problem prob, po_qn qn, po_trsub *out);         /* .h:153-300; qn may be NULL */
int po_trsub_create_eigen(po_problem prob, po_qn eig_qn, po_trsub *out);         /* ...EigenvalueApprox.h:86-206 */
int po_trsub_destroy(po_trsub sub);
int po_trsub_set_eigen_model_update(po_trsub sub, po_eig_update_fn update, void *user); /* setEigenModelUpdate */
/* the subproblem as the ParOptProblem the interior-point solver is built on (borrowed; lives as long as `sub`) */
int po_trsub_problem(po_trsub sub, po_problem *out);
int po_trsub_get_quasi_newton(po_trsub sub, po_qn *qn);                          /* getQuasiNewton (borrowed) */
int po_trsub_init_model_and_bounds(po_trsub sub, double tr_size);
int po_trsub_set_trust_region_bounds(po_trsub sub, double tr_size);
int po_trsub_eval_trial_step_and_update(po_trsub sub, int update_flag, po_vec step, const double *z, po_vec zw,
                                        double *fobj, double *cons);
int po_trsub_accept_trial_step(po_trsub sub, po_vec step, const double *z, po_vec zw);
int po_trsub_reject_trial_step(po_trsub sub);
int po_trsub_get_quasi_newton_update_type(po_trsub sub, int *type);
/* getLinearModel (.h:91-95): returns the number of dense constraints in *m; borrowed pointers */
int po_trsub_get_linear_model(po_trsub sub, po_vec *xk, double *fk, po_vec *gk, const double **ck,
                              const po_vec **Ak, po_vec *lb, po_vec *ub, int *m);
/* A subproblem written by the USER: the virtuals of ParOptTrustRegionSubproblem (src/ParOptTrustRegion.h:15-151) plus the
 * ParOptProblem side the interior point solves (the model in the step s: bounds lk - xk <= s <= uk - xk, model
 * objective and constraints and their gradients; eval_obj_con is also called with step == NULL for the values at
 * s = 0, as the reference's driver does, src/ParOptTrustRegion.cpp:1255).  Every callback returns 0 on success;
 * get_quasi_newton may report NULL (no quasi-Newton term).  get_linear_model hands out BORROWED handles of the
 * user's own vectors (xk, gk, Ak[m], lb, ub) and host values (fk, ck[m]); the library reads it after
 * init_model_and_bounds and after every accept_trial_step.  The sparse constraints (if any) are those of `prob`,
 * linearised about xk by the library as the reference's subproblems do (.h:345-375).  Used by
 * ParOptTrustRegion::optimize / ParOptOptimizer::setTrustRegionSubproblem of the facade for user subclasses and by
 * paropt_amd.TrustRegionSubproblem (Python). */
typedef struct po_trsub_callbacks {
  void *user;
  int (*get_quasi_newton)(void *user, po_qn *qn);
  int (*init_model_and_bounds)(void *user, double tr_size);
  int (*set_trust_region_bounds)(void *user, double tr_size);
  int (*eval_trial_step_and_update)(void *user, int update_flag, po_vec step, const double *z, po_vec zw, double *fobj,
                                    double *cons);
  int (*accept_trial_step)(void *user, po_vec step, const double *z, po_vec zw);
  int (*reject_trial_step)(void *user);
  int (*get_quasi_newton_update_type)(void *user);
  int (*get_linear_model)(void *user, po_vec *xk, double *fk, po_vec *gk, const double **ck, const po_vec **Ak,
                          po_vec *lb, po_vec *ub);
  int (*get_vars_and_bounds)(void *user, po_vec step, po_vec lower, po_vec upper);
  int (*eval_obj_con)(void *user, po_vec step, double *fobj, double *cons);
  int (*eval_obj_con_gradient)(void *user, po_vec step, po_vec g, const po_vec *Ac);
  /* 0: the sparse constraints of `prob` are those of the ORIGINAL problem and are linearised about xk by the library
   * (cw(xk) + Aw(xk) s), as the reference's own subproblems do; non-zero: `prob` is the subproblem's own problem
   * side, whose sparse callbacks already evaluate the model in the step (a facade subclass) */
  int sparse_constraints_are_model;
} po_trsub_callbacks;
int po_trsub_create_callbacks(po_problem prob, const po_trsub_callbacks *callbacks, po_trsub *out);
/* ParOptInfeasSubproblem(subproblem, subproblem_objective, subproblem_constraint) (src/ParOptTrustRegion.h:293-374,
 * .cpp:468-650): the problem the trust-region driver's steering step (minimizeInfeas, .cpp:1105-1228) hands to the
 * interior point, as a po_problem of its own.  The selectors are the reference's constants: objective 1 = the
 * subproblem's model, 2 = the linear model fk + gk^T p, 3 = the constant fk; constraint 1 = the subproblem's model,
 * 2 = the linearisation ck + Ak p.  Bounds, sizes and sparse constraints are the subproblem's.  `sub` is borrowed and
 * must outlive the result; destroy with po_problem_destroy. */
/* a user-written subproblem (po_trsub_create_callbacks) lends its model vectors to the library, which re-reads
 * getLinearModel at initModelAndBounds / acceptTrialStep; call this after changing the model OUTSIDE the trust-region
 * driver (no-op for the library's own subproblems).  po_infeas_create does it once itself. */
int po_trsub_sync_linear_model(po_trsub sub);
int po_infeas_create(po_trsub sub, int subproblem_objective, int subproblem_constraint, po_problem *out);
int po_infeas_set_objective_scaling(po_problem infeas, double scale);             /* setObjectiveScaling .h:309 */
/* ParOptTrustRegion(subproblem, options) and optimize(ip) (src/ParOptTrustRegion.cpp:660-718, 2365-2384).  `ip` must
 * have been created on po_trsub_problem(sub).  Options set on `tr` that the interior-point registry lacks (tr_*,
 * filter_*) are carried into the solver's registry at the call: one registry serves both, as the reference's shared
 * ParOptOptions object does. */
int po_tr_create_subproblem(po_trsub sub, po_tr *out);
int po_tr_optimize_with(po_tr tr, po_ip ip);
int po_tr_initialize(po_tr tr);                                                  /* initialize .cpp:1086-1099 */
int po_tr_set_penalty_gamma(po_tr tr, double gamma);                             /* .cpp:1049-1055 */
int po_tr_set_penalty_gamma_array(po_tr tr, const double *gamma);                /* .cpp:1062-1068 */

/* ---- ParOptMMA: method of moving asymptotes (src/ParOptMMA.h:22-192), assembled as ParOptOptimizer
 * does for algorithm = "mma" (src/ParOptOptimizer.cpp:184-204): the MMA object is the separable
 * rational subproblem handed to an interior-point solver (use_diag_hessian = 1, use_line_search = 0)
 * and the outer loop.  One registry holds the interior-point and the mma_* options
 * (src/ParOptMMA.cpp:234-289).  `prob` is borrowed. */
typedef struct po_mma_s *po_mma;
int po_mma_create(po_problem prob, po_mma *out);
int po_mma_destroy(po_mma mma);
int po_mma_set_option_str(po_mma mma, const char *name, const char *value);
int po_mma_set_option_int(po_mma mma, const char *name, int value);
int po_mma_set_option_float(po_mma mma, const char *name, double value);
int po_mma_optimize(po_mma mma);                             /* optimize .cpp:318-379 */
/* getOptimizedPoint .cpp:489 (+ the multipliers of the last subproblem solve) */
int po_mma_get_optimized_point(po_mma mma, po_vec *x, const double **z, po_vec *zw, po_vec *zl, po_vec *zu);
int po_mma_get_asymptotes(po_mma mma, po_vec *L, po_vec *U); /* getAsymptotes .cpp:494-501 */
/* iteration counters (mma_iter, cumulative subproblem gradient evaluations), objective, constraints */
int po_mma_get_state(po_mma mma, int *mma_iter, int *subproblem_iter, double *fobj, const double **cons);
int po_mma_get_last_row(po_mma mma, const double **row5);    /* fobj, l1, linfty, l1_lambda, infeas */
int po_mma_get_history(po_mma mma, const char **text);       /* the paropt.mma table :584-592 */
typedef int (*po_mma_iteration_fn)(void *user, int mma_iter);
int po_mma_set_iteration_callback(po_mma mma, po_mma_iteration_fn fn, void *user);

#ifdef __cplusplus
}
#endif
#endif /* PAROPT_AMD_H */
