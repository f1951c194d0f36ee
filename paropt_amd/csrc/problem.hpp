// Host-side mirror of the reference's ParOptProblem interface (src/ParOptProblem.h:42-296): dense
// constraints plus the sparse "weighting" constraints with nwblock = 1 (block-diagonal
// Aw D^-1 Aw^T): same method names and argument meaning, device vectors.
#pragma once
#include <vector>

#include "core.hpp"
#include "csr.hpp"
#include "qn.hpp"
#include "wcon.hpp"

namespace po {

class Problem {
 public:
  Problem(Ctx *c, int64_t nlocal_, int ncon_, int nineq_)
      : ctx(c), nlocal(nlocal_), offset(0), nglobal(nlocal_), ncon(ncon_), ninequality(nineq_) {}
  virtual ~Problem();
  virtual int getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) = 0;
  virtual int evalObjCon(Vec *x, double *fobj, double *cons) = 0;
  // Ac == nullptr: only the objective gradient is wanted.  The solver passes nullptr after the first full
  // evaluation of an optimize() call when the problem has declared `linear_constraints` (the dense constraint
  // Jacobian does not depend on x, po_problem_set_linear_constraints), and never otherwise -- the reference's
  // contract (src/ParOptProblem.h:146-158) is unchanged for problems that do not set the flag.
  virtual int evalObjConGradient(Vec *x, Vec *g, Vec **Ac) = 0;
  int linear_constraints = 0;
  // Per-constraint form of the same declaration for the library's own model problems (round 4): entry i != 0 says that
  // the gradient of dense constraint i does not depend on x within one optimize(); the solver then passes Ac[i] ==
  // nullptr after the first evaluation and keeps the column (null: no such knowledge).  Never set for C-ABI problems.
  virtual const std::vector<char> *constantJacobianMask() { return nullptr; }
  virtual int computeQuasiNewtonUpdateCorrection(Vec *x, const double *z, Vec *s, Vec *y) { return 0; }
  // false: the call above leaves s untouched (lets the solver reuse products it already has with the step)
  virtual bool quasiNewtonCorrectionMayChangeStep() { return false; }
  virtual int writeOutput(int iter, Vec *x) { return 0; }
  // true: every reduction the evaluation callbacks issue goes through the internal launchers and tolerates an
  // enclosing BatchScope (core.hpp), so the solver may let them share ONE collective + host sync with its own
  // reductions (trial-point barrier sums with f and c; next residual with the quasi-Newton products).  false
  // (default) for anything that calls out to user code: C-ABI reductions must return their values immediately.
  virtual bool reductionsBatchable() { return false; }
  // ParOptProblem::checkGradients (src/ParOptProblem.cpp:225-622) on device vectors: direction sign(g), forward
  // differences with step dh of the objective and of every dense constraint against g.p and Ac_i.p (and, with
  // check_hvec, of the Lagrangian's gradient against the Hessian-vector product).  work1/work2 are n-sized
  // scratch; the report (one block of text) is appended to *report.  Collective.
  int checkGradients(double dh, Vec *x, bool check_hvec, Vec *work1, Vec *work2, std::string *report);
  virtual int useLowerBounds() { return use_lower; }
  virtual int useUpperBounds() { return use_upper; }
  int use_lower = 1, use_upper = 1;  // setVarBoundOptions of the reference's Cython problems
  // Hessian of the Lagrangian f - z^T c - zw^T cw (src/ParOptProblem.h:160-189); non-zero = not available
  virtual int evalHvecProduct(Vec *x, const double *z, Vec *zw, Vec *px, Vec *hvec) { return 1; }
  virtual int evalHessianDiag(Vec *x, const double *z, Vec *zw, Vec *hdiag) { return 1; }
  // sparse constraints (src/ParOptProblem.h:215-262); out / pzw / A are w-sized device vectors.  The
  // defaults serve a problem with a CSR pattern (`csr`, the ParOptSparseProblem form) and are no-ops
  // otherwise.
  virtual int evalSparseCon(Vec *x, Vec *out);
  virtual int addSparseJacobian(double alpha, Vec *x, Vec *px, Vec *out);
  virtual int addSparseJacobianTranspose(double alpha, Vec *x, Vec *pzw, Vec *out);
  // out <- alpha Aw(x) px (overwrites): by default a zero fill followed by addSparseJacobian; a problem that can write
  // every entry in one pass overrides it
  virtual int setSparseJacobian(double alpha, Vec *x, Vec *px, Vec *out);
  // out <- alpha Aw(x)^T pzw (overwrites): by default a zero fill followed by the call above; a problem that can
  // write every entry in one pass overrides it
  virtual int setSparseJacobianTranspose(double alpha, Vec *x, Vec *pzw, Vec *out);
  // Structured problems describe that n-sized vector instead of writing it (core.hpp: GroupCol): the pass that would
  // have read it forms its entries from pzw in registers.  false (the default): the caller materialises the vector.
  virtual bool sparseTransposeColumn(double alpha, Vec *x, Vec *pzw, GroupCol *col);
  virtual int addSparseInnerProduct(double alpha, Vec *x, Vec *cvec, Vec *A);
  // U_j = Aw (d o P_j) for a whole panel; the default goes column by column through
  // addSparseJacobian with `work` (n-sized) as scratch, structured problems do it in one pass
  virtual int sparseJacobianPanel(Vec *x, Vec *d, const double *const *P, int nv, double *const *U,
                                  Vec *work);
  // Structured problems (one constraint per group of consecutive variables) let that panel image ride in the Gram
  // pass over the same panel (k_wgram's `groups`, one pass over P instead of two): the map and the Jacobian's
  // entry value, or false (the default: the panel image is sparseJacobianPanel's pass of its own)
  virtual bool sparseGramGroups(Vec *x, GramGroups *g);
  // (yx, yw) = K0^-1 (bx, bw) of ParOptQuasiDefBlockMat::apply (src/ParOptSparseMat.cpp:122-190) with
  // the diagonal blocks d (n) and cw (w): yx = d o bx; yw = cw o (bw - Aw yx); yx = d o (bx + Aw^T yw).
  // The default is that sequence through the Jacobian callbacks (11 n-sized passes); structured problems
  // do it in 5.  bx must not alias yx; bw may be null (zero block); `wwork` is w-sized scratch.
  virtual int sparseApplyK0(Vec *x, Vec *d, Vec *cw, const double *bx, const double *bw, Vec *yx, Vec *yw,
                            Vec *wwork);
  // ParOptQuasiDefMat::factor (src/ParOptSparseMat.h:25): cw holds Cdiag on entry.  Block form
  // (nwblock = 1): cw <- 1/(Cdiag + diag(Aw d Aw^T)).  CSR form: S = Cdiag + Aw d Aw^T is factored on the
  // device and cw is left alone.
  virtual int sparseFactor(Vec *x, Vec *d, Vec *cw);
  // The same with Cdiag = sw/zsw + tw/ztw (:1912-1927) formed from the sparse slack blocks on the way (one launch
  // less where the form allows it); the default forms Cdiag into cw and calls sparseFactor
  virtual int sparseFactorFromSlacks(Vec *x, Vec *d, const WVars &v, Vec *cw);
  // Second half of the Gram correction W -= U^T S^-1 U: turns the panel from sparseJacobianPanel into Y
  // with U^T S^-1 U = Y^T diag(weights) Y.  Block form: Y = U, weights = cw.  CSR form: Y = L^-1 U in
  // place, unit weights.
  virtual int sparseHalfSolve(double *const *U, int nv, Vec *cw, const double **weights);
  // out = -S^-1 (U alpha) for the panel as sparseHalfSolve left it (nv columns, w-sized, `out` w-sized):
  // K0^-1 (P alpha, 0) = (Dinv (P alpha + Aw^T out), out), which turns the second quasi-definite apply of a
  // bordered solve into a correction of the first.  Block form: -cw o (U alpha).  CSR form: -L^-T (Y alpha).
  // acc != nullptr: acc += out as well (the caller's running sparse multiplier part), in the same launch where the
  // form allows it
  virtual int sparseCorrection(const double *const *U, int nv, const double *alpha, Vec *cw, Vec *out,
                               Vec *acc = nullptr);
  // one line for the output file, null when there is nothing to say (getFactorInfo, :61)
  virtual const char *sparseFactorInfo();
  // number of factorizations so far that met a non-positive pivot (CSR form; 0 otherwise)
  virtual long sparseFactorBreakdowns() { return csr ? csr->breakdowns : blk_breakdowns; }
  long blk_breakdowns = 0;   // block form with nwblock > 1: factorizations that replaced a non-positive pivot
  int64_t blk_nwcon = 0;     // nwcon the packed-block buffers were sized for
  // fixed CSR pattern of the sparse Jacobian (ParOptSparseProblem::setSparseJacobianData, .cpp:632-677);
  // owned.  Sets nwcon / nwinequality.
  int setSparseJacobianData(int64_t nwcon_, int64_t nwineq_, const int *rowp, const int *cols);
  CsrSparse *csr = nullptr;
  // Structured ("grouped") sparse Jacobian: constraint i acts on the nw consecutive local variables
  // start + i (nw + skip) + [0, nw) and every entry of the Jacobian has the same value group_alpha -- the weighting
  // constraints of multi-material topology optimisation (examples/dmo_truss/dmo_truss_analysis.py:650-679,
  // examples/rosenbrock/rosenbrock.cpp:131-184).  Aw D^-1 Aw^T is then diagonal, the quasi-definite solve is the scalar
  // block form, and every sparse hook below runs a fused group kernel (wcon.hip, the panel image inside the Gram pass,
  // grouped columns formed in registers).  Two ways in (round 5):
  //   * the library's own SeparableProblem::setWeighting (group_alpha = -1), and
  //   * ANY problem that hands its pattern to setSparseJacobianData (the reference's ParOptSparseProblem interface,
  //     src/ParOptProblem.h:301-335): the pattern is RECOGNISED there (rows of equal length nw, consecutive columns,
  //     equal spacing), no symbolic analysis is made, and after every gradient evaluation the entries the user wrote
  //     are checked for equality on the device (one pass over nnz values): differing entries switch the problem to the
  //     general CSR path (analysis + sparse Cholesky) for good.  PAROPT_AMD_NO_CSR_GROUPS=1 switches the recognition off.
  bool grouped = false;
  GroupMap gmap;
  double group_alpha = -1.0;
  // call after the user's evaluation wrote csr->data (CallbackProblem::evalObjConGradient)
  int csrValuesChanged();
  long csr_group_fallbacks = 0;  // times a recognised pattern had to be given up (entries not all equal)
  // block form with nwblock > 1 (ParOptQuasiDefBlockMat, src/ParOptSparseMat.cpp:11-229): consecutive blocks of
  // nwblock constraints may share variables inside a block; addSparseInnerProduct then fills the packed upper
  // triangles of the nwblock x nwblock blocks (nwcon (nwblock+1)/2 entries)
  int setSparseBlockSize(int nwblock_);
  int nwblock = 1;
  Vec *blk = nullptr;    // packed blocks: C + Aw D^-1 Aw^T, then its Cholesky factor U (A = U^T U)
  Vec *wones = nullptr;  // unit weights for the Gram of the half-solved panel
  int *blk_flag = nullptr;
  std::string factor_info;

  Ctx *ctx;
  int64_t nlocal, offset, nglobal;
  int ncon, ninequality;
  int64_t nwcon = 0, nwinequality = 0;  // local counts, as in the reference
};

struct SparseCallbacks {
  po_problem_sparse_callbacks cb;
  bool set = false;
};

// C callback table (the shape of the reference's Cython trampolines, src/CyParOptProblem.h:44-69)
class CallbackProblem : public Problem {
 public:
  CallbackProblem(Ctx *c, int64_t nlocal, int ncon, int nineq, const po_problem_callbacks &cb_)
      : Problem(c, nlocal, ncon, nineq), cb(cb_) {}
  int getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) override;
  int evalObjCon(Vec *x, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *x, Vec *g, Vec **Ac) override;
  int computeQuasiNewtonUpdateCorrection(Vec *x, const double *z, Vec *s, Vec *y) override;
  bool quasiNewtonCorrectionMayChangeStep() override { return cb.qn_update_correction != nullptr; }
  int writeOutput(int iter, Vec *x) override;
  int evalSparseCon(Vec *x, Vec *out) override;
  int addSparseJacobian(double alpha, Vec *x, Vec *px, Vec *out) override;
  int addSparseJacobianTranspose(double alpha, Vec *x, Vec *pzw, Vec *out) override;
  int addSparseInnerProduct(double alpha, Vec *x, Vec *cvec, Vec *A) override;
  int evalHvecProduct(Vec *x, const double *z, Vec *zw, Vec *px, Vec *hvec) override;
  int evalHessianDiag(Vec *x, const double *z, Vec *zw, Vec *hdiag) override;
  // po_problem_set_deferred_reductions: the problem's callbacks obtain reduced values only through this library's
  // reductions (po_vec_dot/mdot/..., po_ctx_reduce_device) and post-process them in po_ctx_after_reduce hooks, so
  // the solver may let them share a collective + host sync with its own (reductionsBatchable, above)
  int deferred_reductions = 0;
  bool reductionsBatchable() override { return deferred_reductions != 0; }
  po_problem_callbacks cb;
  SparseCallbacks sparse;
  // CSR form (CyParOptSparseProblem, src/CyParOptProblem.h:177-262): the two evaluation callbacks also fill
  // the sparse constraint values and the Jacobian entries
  po_eval_sparse_obj_con_fn csr_obj_con = nullptr;
  po_eval_sparse_obj_con_gradient_fn csr_gradient = nullptr;
  po_hvec_fn hvec_fn = nullptr;
  po_hdiag_fn hdiag_fn = nullptr;
};

// Device-resident separable workloads (DESIGN.md "Workloads").
class SeparableProblem : public Problem {
 public:
  SeparableProblem(Ctx *c, int kind, int64_t nglobal, int ncon, uint64_t seed, double eig_min,
                   double eig_max);
  ~SeparableProblem();
  int init();
  int getVarsAndBounds(Vec *x, Vec *lb, Vec *ub) override;
  int evalObjCon(Vec *x, double *fobj, double *cons) override;
  int evalObjConGradient(Vec *x, Vec *g, Vec **Ac) override;
  bool reductionsBatchable() override { return true; }
  double rosen_out[3] = {0, 0, 0};  // landing area of the Rosenbrock reductions (BatchScope::end_then)
  // weighting constraints cw_i = 1 - sum_{k<nw} x[nwstart + i (nw + nwskip) + k] on GLOBAL indices;
  // groups must not straddle rank boundaries (checked)
  int setWeighting(int64_t nwcon_global, int nw, int64_t nwstart, int nwskip, int64_t nwineq_global);
  // overlapping nonlinear sparse constraints in CSR form, rank-local like the reference's example
  // (examples/rosenbrock/sparse_rosenbrock.cpp:38-118 is span = 2, stride = 1):
  //   cw_i = 1 - sum_{k<span} x[i*stride + k]^2,  i < (nlocal - span)/stride + 1, all inequalities
  int setChain(int span, int stride, int reverse_cols);
  int chain_span = 0, chain_stride = 0, chain_reverse = 0;
  Vec *chain_tmp = nullptr;
  int chainHessian(Vec *zw, Vec *px, Vec *h);
  // (the Jacobian hooks of the weighting constraints are the base class's grouped forms)
  int evalSparseCon(Vec *x, Vec *out) override;
  int evalHvecProduct(Vec *x, const double *z, Vec *zw, Vec *px, Vec *hvec) override;
  int evalHessianDiag(Vec *x, const double *z, Vec *zw, Vec *hdiag) override;
  int bounds_mode = 0;  // po_problem_set_bounds_mode: deliberately broken bounds (k_bounds_mode)

  int kind;
  uint64_t seed;
  double eig_min, eig_max;
  Vec *q, *b;             // objective data
  std::vector<Vec *> A;   // constraint rows a_j (device)
  std::vector<double> beta;
};

}  // namespace po

struct po_problem_s {
  po::Problem *p;
};
