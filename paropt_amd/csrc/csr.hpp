// The general sparse-constraint path: a fixed CSR pattern for the Jacobian Aw of the rank-local sparse
// constraints (ParOptSparseProblem, reference src/ParOptProblem.h:301-395, src/ParOptProblem.cpp:624-816)
// and the quasi-definite solve
//     [ D   Aw^T ] [ yx ]   [ bx ]
//     [ Aw  -C   ] [-yw ] = [ bw ]          S = C + Aw D^-1 Aw^T   sparse, SPD
// of ParOptQuasiDefSparseMat (src/ParOptSparseMat.cpp:234-450).
//
// Split of the work (MI355X first, not the reference's host supernodal code):
//   host, ONCE per pattern : sorted CSR/CSC, pattern of S, nested-dissection ordering from BFS level
//                            structures, elimination tree, pattern of L, dependency level sets
//   device, every iteration: S assembled straight into L's value array (one thread per structural
//                            entry), a level-scheduled row Cholesky (one wavefront per row), level-scheduled
//                            triangular solves for a whole panel of right-hand sides, CSR/CSC products
// Everything is rank-local (the reference's sparse constraints never couple ranks), deterministic
// (no atomics), and there is no host numeric path.
//
// PARITY UNPINNED for the factorization itself: the reference's ParOptSparseCholesky.cpp needs METIS,
// which this image lacks, so no reference-run golden exists for it.  What IS pinned: trajectories of
// the reference's interior point on ParOptSparseProblem (its own CSR products) with a dense LAPACK
// S-solve supplied by oracle/ref_driver.cpp through createQuasiDefMat() (tests/golden/ipcsr_*.npz); a
// direct solve is unique up to round-off, and tests also compare against a dense numpy Cholesky.
#pragma once
#include <string>
#include <vector>

#include "core.hpp"
#include "qn.hpp"

namespace po {

// host result of the one-time analysis (all indices int32; sizes checked)
struct CsrSymbolic {
  int w = 0, n = 0;
  int64_t nnz = 0;
  std::vector<int> rowp, cols, src;     // A sorted by column within each row; src = user slot of each entry
  std::vector<int> colp, rowsT, srcT;   // A^T: srcT indexes the SORTED value array
  std::vector<int> perm, iperm;         // perm[new] = old
  std::vector<int> parent;              // elimination tree of P S P^T
  std::vector<int> Lrowp, Lcols;        // rows of L (lower, columns ascending, diagonal last)
  std::vector<int> Ltp, Ltrows, Ltsrc;  // columns of L below the diagonal, Ltsrc = slot in the row storage
  std::vector<int> ent_a, ent_b, ent_slot;  // structural entries of lower(P S P^T): rows a,b of A -> slot in L
  // rows grouped by dependency level: ascending for the factorization and the forward solve, descending for
  // the backward solve.  Rows are NUMBERED level by level, so fwd_order is the identity and each level is a
  // contiguous range of rows (and of L's storage).
  std::vector<int> fwd_order, fwd_ptr;
  std::vector<int> fwd_maxlen, bwd_maxlen;  // longest ordinary row / column of each level (kernel variant choice)
  // Each level = ordinary rows [fwd_ptr[l], ord_end[l]) followed by the rows of its FRONTS (dense separator
  // cliques, front_ptr[l] .. front_ptr[l+1]-1 in front_start / front_size): row front_start + r ends with the
  // columns front_start .. front_start + r, so the front's lower triangle is the tails of its rows.
  std::vector<int> ord_end, front_ptr, front_start, front_size, front_of;
  std::vector<int> front_maxdesc;  // per level: longest part of a front row left of its front
  int max_front = 0;
  int64_t nnzS = 0, nnzL = 0;
  bool identity_src = true;
};

// returns PO_OK or PO_ERR_ARG (message set); pure host code, testable without a GPU
int csr_analyse(int64_t n, int64_t w, const int *rowp, const int *cols, CsrSymbolic *out);

class CsrSparse {
 public:
  CsrSparse(Ctx *c, int64_t n_, int64_t w_);
  ~CsrSparse();
  int setPattern(const int *rowp, const int *cols);
  // The pattern was recognised as the grouped one (problem.hpp): only what the user's callbacks and
  // getSparseJacobianData need -- host copies of the pattern, the value array, the constraint vector -- and NO
  // symbolic analysis (at w = 1 M rows it would cost seconds and hundreds of MB that the block form never uses).
  // upgradeFromLight() runs the analysis after all, keeping the values (the entries turned out not to be uniform).
  int setPatternLight(const int *rowp, const int *cols);
  int upgradeFromLight();
  bool light = false;
  double *adopt_data = nullptr;  // upgradeFromLight -> setPattern: keep these allocations instead of new ones
  Vec *adopt_cw = nullptr;
  // the user's value array in the user's entry order (device, nnz doubles) and the constraint values
  double *data = nullptr;
  Vec *cw = nullptr;
  // call after `data` changed (evalObjConGradient): refreshes the column-sorted copy
  int valuesChanged();
  int spmv(double alpha, const double *px, double *out);     // out += alpha Aw px        (w)
  int spmvT(double alpha, const double *pzw, double *out);   // out += alpha Aw^T pzw     (n)
  int innerProduct(double alpha, const double *cvec, double *out);  // out_i += alpha sum_k a_ik^2 c_k
  int colSum(double scale, const double *y, double *out);    // out_j = scale * sum of y over the rows touching j
  // U_j = Aw (d o P_j), rows in the factor's elimination order (only the Gram correction reads them)
  int panelPermuted(const double *d, const double *const *P, int nv, double *const *U);
  int factor(const double *dinv, const double *cdiag);       // S = diag(cdiag) + Aw diag(dinv) Aw^T = L L^T
  int halfSolve(double *const *U, int nv);                   // U_j <- L^-1 U_j (elimination order, in place)
  // out (natural order) = -S^-1 (U alpha) from the half-solved panel Y = L^-1 U: -L^-T (Y alpha)
  int correction(const double *const *Y, int nv, const double *alpha, double *out);
  // (yx, yw) = K0^-1 (bx, bw); bw may be null; bx must not alias yx
  int applyK0(const double *dinv, const double *bx, const double *bw, double *yx, double *yw);
  const double *unitWeights() const { return ones; }
  const char *factorInfo();
  // host copies for getSparseJacobianData (src/ParOptProblem.cpp:689-703)
  std::vector<int> user_rowp, user_cols;
  CsrSymbolic sym;
  Ctx *ctx;
  int64_t n, w, nnz;
  int nlevels_f = 0;
  long breakdowns = 0;  // factorizations that met a non-positive pivot (see factor())

 private:
  int solveInPlace(double *const *Y, int nv, bool forward, bool backward);
  double *vals = nullptr;  // column-sorted values (== data when the user's order is already sorted)
  int *d_rowp = nullptr, *d_cols = nullptr, *d_src = nullptr;
  int *d_colp = nullptr, *d_rowsT = nullptr, *d_srcT = nullptr;
  int *d_perm = nullptr, *d_iperm = nullptr;
  int *d_Lrowp = nullptr, *d_Lcols = nullptr;
  int *d_Ltp = nullptr, *d_Ltrows = nullptr, *d_Ltsrc = nullptr;
  int *d_ent_a = nullptr, *d_ent_b = nullptr, *d_ent_slot = nullptr;
  int *d_fwd = nullptr;
  int *d_front_of = nullptr, *d_front_end = nullptr, *d_fstart = nullptr, *d_fsize = nullptr;
  double *Lvals = nullptr, *ones = nullptr, *wwork = nullptr;
  int *d_flag = nullptr;
  std::string info;
  int spmv_group = 1, spmvT_group = 1;
};

// kernels (csr.hip)
int k_csr_gather(Ctx *c, double *dst, const double *src, const int *idx, int64_t n);
int k_csr_spmv(Ctx *c, int group, const int *rowp, const int *cols, const double *vals, int64_t w, double alpha,
               const double *x, const double *scale, double beta, const double *b, double *out, const int *outperm);
int k_csr_spmvT(Ctx *c, int group, const int *colp, const int *rowsT, const int *srcT, const double *vals,
                int64_t n, double alpha, const double *y, const double *bx, const double *dscale, double *out);
int k_csr_colsum(Ctx *c, const int *colp, const int *rowsT, int64_t n, double scale, const double *y, double *out);
int k_csr_inner(Ctx *c, const int *rowp, const int *cols, const double *vals, int64_t w, double alpha,
                const double *cvec, double *out);
int k_csr_panel(Ctx *c, const int *rowp, const int *cols, const double *vals, int64_t w, const double *d,
                const double *const *P, int nv, double *const *U, const int *outperm);
int k_csr_assemble(Ctx *c, const int *rowp, const int *cols, const double *vals, const double *dinv,
                   const double *cdiag, const int *ent_a, const int *ent_b, const int *ent_slot, int64_t nent,
                   double *Lvals);
int k_chol_level(Ctx *c, const int *Lrowp, const int *Lcols, double *Lvals, const int *rows, int nrows, int *flag,
                 int thin, int maxlen);
int k_chol_fronts(Ctx *c, const int *Lrowp, const int *Lcols, double *Lvals, int row0, int nrows,
                  const int *front_of, const int *fstart, const int *fsize, int nfronts, int maxdesc, int *flag);
int k_trsv_fronts_fwd(Ctx *c, const int *Lrowp, const int *Lcols, const double *Lvals, int row0, int nrows,
                      const int *front_of, const int *fstart, const int *fsize, int nfronts, double *const *Y,
                      int nv);
int k_trsv_fronts_bwd(Ctx *c, const int *Lrowp, const int *Ltp, const int *Ltrows, const int *Ltsrc,
                      const double *Lvals, int row0, int nrows, const int *front_of, const int *front_end,
                      const int *fstart, const int *fsize, int nfronts, double *const *Y, int nv);
int k_trsv_fwd_level(Ctx *c, const int *Lrowp, const int *Lcols, const double *Lvals, const int *rows, int nrows,
                     double *const *Y, int nv, int thin);
int k_trsv_bwd_level(Ctx *c, const int *Lrowp, const int *Ltp, const int *Ltrows, const int *Ltsrc,
                     const double *Lvals, const int *rows, int nrows, double *const *Y, int nv, int thin);
// built-in chain constraints cw_i = 1 - sum_{k<span} x[i*stride + k]^2 and their Jacobian entries
int k_chain_con(Ctx *c, const double *x, int64_t w, int span, int stride, double *cw);
int k_chain_jac(Ctx *c, const double *x, int64_t w, int span, int stride, int reverse, double *data);

}  // namespace po
