// Compact limited-memory quasi-Newton approximations with the S/Y/Z panels resident in HBM.
// Host-side mirror of ParOptCompactQuasiNewton / ParOptLBFGS / ParOptLSR1
// (reference src/ParOptQuasiNewton.h:32-220): same method names, argument meaning and return
// codes; the k x k matrices (B = S^T S, L, D, M and its LU factors) stay on the host, replicated
// on every rank, exactly as in the reference.
#pragma once
#include <vector>

#include "core.hpp"

namespace po {

Vec *vec_new(Ctx *c, int64_t n);
void vec_decref(Vec *v);
void live_objects(long *vecs, long long *bytes);
void mirror_created();
void mirror_freed();
long live_mirrors();

class CompactQuasiNewton {
 public:
  CompactQuasiNewton(Ctx *ctx, int64_t n, int msub_max, bool keep_z);
  virtual ~CompactQuasiNewton();

  void setInitDiagonalType(int t) { diag_type = t; }
  virtual void reset();
  // returns PO_* status; *rc = 0 normal, 1 damped, 2 skipped (src/ParOptQuasiNewton.cpp:162-334)
  virtual int update(Vec *s, Vec *y, int *rc) = 0;
  // The same with Z^T s already known to the caller (the interior point has P^T px of the step it is about to
  // store: s = alpha px): the pass over the whole panel inside update() shrinks to the columns that are not
  // implied -- none for L-BFGS (Z = [S | Y]), the S columns for L-SR1 (Y_j.s = Z_j.s + b0 S_j.s).  zTs has
  // size() entries in the order of the current panel.  Default: ignores the hint.
  virtual int updateWithZTs(Vec *s, Vec *y, const double *zTs, int *rc) { return update(s, y, rc); }
  // true: update() issues its reductions through the internal launchers only and tolerates an enclosing BatchScope
  // (core.hpp); false for approximations that call out to user code (ParOptEigenQuasiNewton's update callback)
  virtual bool reductionsBatchable() const { return false; }
  // Set by a caller that owns s and y and rewrites them completely before their next use (the interior point's
  // s_qn / y_qn): the update then takes their device buffers into the pair slots and hands the slots' old buffers
  // back instead of copying 2 n-vectors (the copies of src/ParOptQuasiNewton.cpp:266-303).  s and y hold
  // unspecified values afterwards.  Never set for vectors that belong to user code.
  bool take_buffers = false;
  // update(x, z, zw): multiplier-only update, a no-op for the limited-memory classes
  // (src/ParOptQuasiNewton.h:60-63); ParOptEigenQuasiNewton records z[index] here
  virtual int updateMult(Vec *x, const double *z, Vec *zw) { return 0; }
  virtual int mult(Vec *x, Vec *y);                    // y = B x
  virtual int multAdd(double alpha, Vec *x, Vec *y);   // y += alpha B x
  // (b0, d0, M, Z) of B = b0 I - Z diag(d0) M^-1 diag(d0) Z^T; returns the size k
  virtual int getCompactMat(double *b0_, const double **d0_, const double **M_, Vec ***Z_);
  virtual int getMaxLimitedMemorySize() = 0;

  // rz <- diag(d0) M^-1 diag(d0) rz   (the host part of mult, :399-411)
  virtual void applyCompactInverse(double *rz) const;
  virtual int size() const { return (int)Z.size(); }
  virtual double diag() const { return b0; }
  virtual std::vector<const double *> zPointers() const;
  // L-SR1 only: after update() the columns Z_j = Y_j - b0 S_j are left unformed until somebody needs
  // them.  A consumer that can form them on the fly (the weighted-Gram pass of the interior point)
  // asks for the ingredients and then declares them materialised; everybody else goes through
  // zPointers() / getCompactMat() / mult(), which form them first (one fused 3-pass launch).
  virtual bool pendingZ(std::vector<const double *> *Yp, std::vector<const double *> *Sp,
                        std::vector<double *> *Zout, double *b0_) const {
    return false;
  }
  virtual void pendingZDone() {}
  virtual int ensureZ() const { return PO_OK; }

  // Test hook (po_qn_debug_load, state-injected known-answer tests): takes over a complete limited-memory state --
  // msub_ pairs, b0 and the small matrices the updates maintain (B = S^T S, the strictly lower triangle L of S^T Y,
  // D = diag(S^T Y); column-major with leading dimension ld, as src/ParOptQuasiNewton.h:141-147 holds them) -- and
  // rebuilds (d0, M, its LU factors, Z) from it with the arithmetic of update().  L-SR1 columns are left unformed,
  // as after an update.
  int debugLoad(int msub_, double b0_, const double *B_, const double *L_, const double *D_, int ld, Vec *const *S_,
                Vec *const *Y_);

  Ctx *ctx;
  int64_t n;

 protected:
  virtual void rebuildCompact() {}  // (d0, M, LU, Z) from (b0, B, L, D, msub): the tail of update()
  // append / rotate a pair and refresh B, D, L from dots already available on the host:
  // sS[i] = s.S_i, sY[i] = s.Y_i for the pairs held BEFORE the update (old ordering)
  int storePair(Vec *s, Vec *y, const double *sS, const double *sY, double sTs, double sTy);
  void factorM();

 public:
  const std::vector<int> &pivots() const { return piv; }  // 0-based LU pivot rows of M (po_qn_get_pivots)

 protected:

  int msub_max, msub;
  double b0;
  int diag_type;
  std::vector<Vec *> S, Y;  // owned, in current (rotated) order
  std::vector<Vec *> Zown;  // L-SR1 only: materialised Z_i = Y_i - b0 S_i
  std::vector<Vec *> Z;     // borrowed: current compact panel
  Vec *r;                   // scratch (damped update)
  std::vector<double> D, L, B;  // msub_max, msub_max^2 (column-major, ld = msub_max)
  std::vector<double> M, Mf, d0;
  std::vector<int> piv;
};

class LBFGS : public CompactQuasiNewton {
 public:
  LBFGS(Ctx *ctx, int64_t n, int msub_max) : CompactQuasiNewton(ctx, n, msub_max, false) {
    update_type = PO_BFGS_SKIP_NEGATIVE_CURVATURE;
  }
  void setBFGSUpdateType(int t) { update_type = t; }
  int update(Vec *s, Vec *y, int *rc) override { return updateWithZTs(s, y, nullptr, rc); }
  int updateWithZTs(Vec *s, Vec *y, const double *zTs, int *rc) override;
  bool reductionsBatchable() const override { return true; }
  int getMaxLimitedMemorySize() override { return 2 * msub_max; }

 protected:
  void rebuildCompact() override { computeMatUpdate(); }

 private:
  void computeMatUpdate();
  int update_type;
};

class LSR1 : public CompactQuasiNewton {
 public:
  LSR1(Ctx *ctx, int64_t n, int msub_max) : CompactQuasiNewton(ctx, n, msub_max, true), z_pending(false) {}
  void reset() override {
    z_pending = false;
    CompactQuasiNewton::reset();
  }
  int update(Vec *s, Vec *y, int *rc) override { return updateWithZTs(s, y, nullptr, rc); }
  int updateWithZTs(Vec *s, Vec *y, const double *zTs, int *rc) override;
  bool reductionsBatchable() const override { return true; }
  int getMaxLimitedMemorySize() override { return msub_max; }
  bool pendingZ(std::vector<const double *> *Yp, std::vector<const double *> *Sp, std::vector<double *> *Zout,
                double *b0_) const override;
  void pendingZDone() override { z_pending = false; }
  int ensureZ() const override;

 protected:
  void rebuildCompact() override;

 private:
  mutable bool z_pending;
};

}  // namespace po

struct po_qn_s {
  po::CompactQuasiNewton *qn;
  std::vector<po_vec> zhandles;
};
