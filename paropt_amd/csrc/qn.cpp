// Compact L-BFGS / L-SR1 on device panels (see qn.hpp).
//
// Traffic per call (doubles / n): mult = 2k+3 (one mdot pass + one panel-axpy pass) instead of
// the reference's 4+5k; update = k+5 (+ 2k+3 when damped) instead of (18+7k): the dots
// s.S_i, s.Y_i that the reference recomputes for the new Gram row (:307-321) are exactly the
// Z^T s it already formed inside mult(s) (:183), so they are reused.
#include "qn.hpp"

#include <math.h>
#include <string.h>

#include <atomic>

namespace po {

// Device vectors are allocated with an even element count plus slack and zero-filled, so the
// kernels can always use full 16-byte accesses on the last pair (kernels.hip::ld2/st2).
static size_t padded_elems(int64_t n) { return (size_t)(((n + 1) >> 1) << 1) + 2; }

// live-object accounting (po_live_objects): every device vector alive and the bytes behind them
static std::atomic<long> g_live_vecs(0);
static std::atomic<long long> g_live_bytes(0);
static std::atomic<long> g_live_mirrors(0);  // pinned host mirrors behind po_vec_get_array
void mirror_created() { g_live_mirrors++; }
void mirror_freed() { g_live_mirrors--; }
long live_mirrors() { return g_live_mirrors.load(); }
void live_objects(long *vecs, long long *bytes) {
  if (vecs) *vecs = g_live_vecs.load();
  if (bytes) *bytes = g_live_bytes.load();
}

Vec *vec_new(Ctx *c, int64_t n) {
  po_vec_s *v = new po_vec_s();
  v->ctx = c;
  v->n = n;
  v->d = nullptr;
  v->ref = 1;
  v->h = nullptr;
  const size_t bytes = sizeof(double) * padded_elems(n);
  if (hipSetDevice(c->device) != hipSuccess || hipMalloc((void **)&v->d, bytes) != hipSuccess) {
    set_error("hipMalloc of %lld doubles failed", (long long)n);
    delete v;
    return nullptr;
  }
  if (hipMemsetAsync(v->d, 0, bytes, c->stream) != hipSuccess) {
    set_error("hipMemsetAsync failed");
  }
  g_live_vecs++;
  g_live_bytes += (long long)bytes;
  return v;
}

void vec_decref(Vec *v) {
  if (!v) return;
  if (--v->ref == 0) {
    (void)hipStreamSynchronize(v->ctx->stream);
    if (v->d) (void)hipFree(v->d);
    if (v->h) {
      (void)hipHostFree(v->h);
      mirror_freed();
    }
    g_live_vecs--;
    g_live_bytes -= (long long)(sizeof(double) * padded_elems(v->n));
    delete static_cast<po_vec_s *>(v);
  }
}

CompactQuasiNewton::CompactQuasiNewton(Ctx *ctx_, int64_t n_, int msub_max_, bool keep_z)
    : ctx(ctx_), n(n_), msub_max(msub_max_), msub(0), b0(1.0), diag_type(PO_QN_YTY_OVER_YTS) {
  for (int i = 0; i < msub_max; i++) {
    S.push_back(vec_new(ctx, n));
    Y.push_back(vec_new(ctx, n));
    if (keep_z) Zown.push_back(vec_new(ctx, n));
  }
  r = vec_new(ctx, n);
  D.assign(msub_max, 0.0);
  L.assign((size_t)msub_max * msub_max, 0.0);
  B.assign((size_t)msub_max * msub_max, 0.0);
}

CompactQuasiNewton::~CompactQuasiNewton() {
  for (Vec *v : S) vec_decref(v);
  for (Vec *v : Y) vec_decref(v);
  for (Vec *v : Zown) vec_decref(v);
  vec_decref(r);
}

void CompactQuasiNewton::reset() {  // :127-142 / :603-618
  msub = 0;
  b0 = 1.0;
  std::fill(D.begin(), D.end(), 0.0);
  std::fill(L.begin(), L.end(), 0.0);
  std::fill(B.begin(), B.end(), 0.0);
  M.clear();
  Mf.clear();
  d0.clear();
  piv.clear();
  Z.clear();
}

std::vector<const double *> CompactQuasiNewton::zPointers() const {
  ensureZ();
  std::vector<const double *> p;
  for (Vec *v : Z) p.push_back(v->d);
  return p;
}

int CompactQuasiNewton::getCompactMat(double *b0_, const double **d0_, const double **M_, Vec ***Z_) {
  if (Z_) ensureZ();
  if (b0_) *b0_ = b0;
  if (d0_) *d0_ = d0.data();
  if (M_) *M_ = M.data();
  if (Z_) *Z_ = Z.data();
  return (int)Z.size();
}

void CompactQuasiNewton::applyCompactInverse(double *rz) const {
  const int k = (int)Z.size();
  if (k == 0) return;
  for (int i = 0; i < k; i++) rz[i] *= d0[i];
  lu_solve(k, Mf.data(), k, piv.data(), rz);
  for (int i = 0; i < k; i++) rz[i] *= d0[i];
}

void CompactQuasiNewton::factorM() {
  const int k = (int)d0.size();
  Mf = M;
  piv.assign(k, 0);
  if (k > 0) lu_factor(k, Mf.data(), k, piv.data());
}

int CompactQuasiNewton::mult(Vec *x, Vec *y) {
  const int k = (int)Z.size();
  std::vector<double> rz(k > 0 ? k : 1, 0.0);
  std::vector<const double *> zp = zPointers();
  if (k > 0) {
    PO_TRY(k_mdot(ctx, x->d, zp.data(), k, n, rz.data()));
    applyCompactInverse(rz.data());
    for (int i = 0; i < k; i++) rz[i] = -rz[i];
  }
  return k_panel_axpy(ctx, y->d, b0, x->d, 0.0, rz.data(), zp.data(), k, n);
}

int CompactQuasiNewton::multAdd(double alpha, Vec *x, Vec *y) {
  const int k = (int)Z.size();
  std::vector<double> rz(k > 0 ? k : 1, 0.0);
  std::vector<const double *> zp = zPointers();
  if (k > 0) {
    PO_TRY(k_mdot(ctx, x->d, zp.data(), k, n, rz.data()));
    applyCompactInverse(rz.data());
    for (int i = 0; i < k; i++) rz[i] = -alpha * rz[i];
  }
  return k_panel_axpy(ctx, y->d, alpha * b0, x->d, 1.0, rz.data(), zp.data(), k, n);
}

int CompactQuasiNewton::storePair(Vec *s, Vec *y, const double *sS, const double *sY, double sTs,
                                  double sTy) {
  // pointer rotation and matrix shifts: src/ParOptQuasiNewton.cpp:266-303
  int shift = 0;
  // the pair enters its slot by buffer exchange when the caller allows it (take_buffers), by copy otherwise
  auto put = [&](Vec *slot, Vec *src) -> int {
    if (take_buffers && src->n == slot->n && !src->h_live && !slot->h_live) {
      std::swap(slot->d, src->d);
      return PO_OK;
    }
    return k_copy(ctx, slot->d, src->d, n);
  };
  if (msub < msub_max) {
    PO_TRY(put(S[msub], s));
    PO_TRY(put(Y[msub], y));
    msub++;
  } else if (msub == msub_max && msub_max > 0) {
    shift = 1;
    PO_TRY(put(S[0], s));
    PO_TRY(put(Y[0], y));
    Vec *st = S[0], *yt = Y[0];
    for (int i = 0; i < msub - 1; i++) {
      S[i] = S[i + 1];
      Y[i] = Y[i + 1];
    }
    S[msub - 1] = st;
    Y[msub - 1] = yt;
    if (!Zown.empty()) {
      Vec *zt = Zown[0];
      for (int i = 0; i < msub - 1; i++) Zown[i] = Zown[i + 1];
      Zown[msub - 1] = zt;
    }
    const int m = msub_max;
    for (int i = 0; i < msub - 1; i++) D[i] = D[i + 1];
    for (int i = 0; i < msub - 1; i++)
      for (int j = 0; j < msub - 1; j++) B[i + (size_t)j * m] = B[i + 1 + (size_t)(j + 1) * m];
    for (int i = 0; i < msub - 1; i++)
      for (int j = 0; j < i; j++) L[i + (size_t)j * m] = L[i + 1 + (size_t)(j + 1) * m];
  } else {
    return PO_OK;  // msub_max == 0
  }
  // new Gram row from the dots with the previously held pairs (:307-321)
  const int m = msub_max, k = msub;
  for (int i = 0; i < k - 1; i++) {
    const double v = sS[i + shift];
    B[(k - 1) + (size_t)i * m] = v;
    B[i + (size_t)(k - 1) * m] = v;
  }
  B[(k - 1) + (size_t)(k - 1) * m] = sTs;
  D[k - 1] = sTy;
  for (int i = 0; i < k - 1; i++) L[(k - 1) + (size_t)i * m] = sY[i + shift];
  return PO_OK;
}

int CompactQuasiNewton::debugLoad(int msub_, double b0_, const double *B_, const double *L_, const double *D_, int ld,
                                  Vec *const *S_, Vec *const *Y_) {
  if (msub_ < 0 || msub_ > msub_max || ld < msub_) {
    set_error("po_qn_debug_load: %d pairs do not fit the subspace of %d (ld %d)", msub_, msub_max, ld);
    return PO_ERR_ARG;
  }
  reset();
  msub = msub_;
  b0 = b0_;
  const int m = msub_max;
  for (int j = 0; j < msub; j++) {
    if (!S_[j] || !Y_[j] || S_[j]->n != n || Y_[j]->n != n) {
      set_error("po_qn_debug_load: pair %d does not have the local size %lld", j, (long long)n);
      return PO_ERR_ARG;
    }
    PO_TRY(k_copy(ctx, S[j]->d, S_[j]->d, n));
    PO_TRY(k_copy(ctx, Y[j]->d, Y_[j]->d, n));
    D[j] = D_[j];
    for (int i = 0; i < msub; i++) {
      B[i + (size_t)m * j] = B_[i + (size_t)ld * j];
      L[i + (size_t)m * j] = L_[i + (size_t)ld * j];
    }
  }
  if (msub > 0) rebuildCompact();
  return PO_OK;
}

// ------------------------------------------------------------------------------------------------
void LBFGS::computeMatUpdate() {  // :339-377
  const int k = msub, m = msub_max;
  M.assign((size_t)4 * k * k, 0.0);
  for (int i = 0; i < k; i++)
    for (int j = 0; j < k; j++) M[i + (size_t)2 * k * j] = b0 * B[i + (size_t)m * j];
  for (int i = 0; i < k; i++) {
    for (int j = 0; j < i; j++) {
      M[i + (size_t)2 * k * (j + k)] = L[i + (size_t)m * j];
      M[j + k + (size_t)2 * k * i] = L[i + (size_t)m * j];
    }
  }
  for (int i = 0; i < k; i++) M[k + i + (size_t)2 * k * (k + i)] = -D[i];
  d0.assign(2 * k, 1.0);
  for (int i = 0; i < k; i++) d0[i] = b0;
  Z.clear();
  for (int i = 0; i < k; i++) Z.push_back(S[i]);
  for (int i = 0; i < k; i++) Z.push_back(Y[i]);
  factorM();
}

int LBFGS::updateWithZTs(Vec *s, Vec *y, const double *zTs, int *rc) {
  *rc = 0;
  const int k = (int)Z.size();  // = 2*msub of the current panel
  const int mold = k / 2;
  // one pass: [Z^T s, s.s, s.y] (Z^T s from the caller when it has it); one more reduction for y.y
  std::vector<double> dots(k + 2, 0.0);
  BatchScope batch(ctx);  // both reductions (and whatever the caller has queued) share one collective + sync
  if (zTs) {
    const double *two[2] = {s->d, y->d};
    PO_TRY(k_mdot(ctx, s->d, two, 2, n, dots.data() + k));
    for (int i = 0; i < k; i++) dots[i] = zTs[i];
  } else {
    std::vector<const double *> vp = zPointers();
    vp.push_back(s->d);
    vp.push_back(y->d);
    PO_TRY(k_mdot(ctx, s->d, vp.data(), k + 2, n, dots.data()));
  }
  double yTy = 0.0;
  PO_TRY(k_reduce1(ctx, RED_SUMSQ, y->d, nullptr, n, &yTy));
  PO_TRY(batch.end());
  double sTs = dots[k], yTs = dots[k + 1];
  if (1e-8 * yTy >= fabs(yTs)) {  // Nocedal skip :175-179
    *rc = 2;
    return PO_OK;
  }
  // s^T B s = b0 s.s - rz^T diag(d0) M^-1 diag(d0) rz   (mult(s) + dot of :183-186)
  std::vector<double> coef(dots.begin(), dots.begin() + k);
  applyCompactInverse(coef.data());
  double sTBs = b0 * sTs;
  for (int i = 0; i < k; i++) sTBs -= dots[i] * coef[i];

  const double epsilon_precision = 1e-12;
  double b0_init = b0;
  if (yTs >= epsilon_precision) {
    b0_init = (diag_type == PO_QN_YTS_OVER_STS) ? yTs / sTs : yTy / yTs;
  } else {
    b0_init = 0.5 * (fabs(yTy / yTs) + fabs(yTs / sTs));
  }
  Vec *y_update = nullptr;
  if (yTs >= 0.01 * sTBs) {
    y_update = y;
    b0 = b0_init;
  } else if (update_type == PO_BFGS_SKIP_NEGATIVE_CURVATURE) {
    *rc = 2;
    return PO_OK;
  } else {  // damped update :241-263
    *rc = 1;
    const double theta = 0.8 * sTBs / (sTBs - yTs);
    // r = (1-theta) B s + theta y, with B s = b0 s - Z coef
    std::vector<double> alpha(k + 1);
    for (int i = 0; i < k; i++) alpha[i] = -(1.0 - theta) * coef[i];
    alpha[k] = theta;
    std::vector<const double *> ap = zPointers();
    ap.push_back(y->d);
    PO_TRY(k_panel_axpy(ctx, r->d, (1.0 - theta) * b0, s->d, 0.0, alpha.data(), ap.data(), k + 1, n));
    y_update = r;
    const double *two[2] = {r->d, s->d};
    double d2[2];
    BatchScope damped(ctx);  // the values are needed now, also when the caller has a batch open
    PO_TRY(k_mdot(ctx, r->d, two, 2, n, d2));
    PO_TRY(damped.end());
    yTy = d2[0];
    yTs = d2[1];
    b0 = (diag_type == PO_QN_YTS_OVER_STS) ? yTs / sTs : yTy / yTs;
  }
  // the L row needs s.Y_i of the held pairs: rows mold..2mold-1 of Z^T s
  PO_TRY(storePair(s, y_update, dots.data(), dots.data() + mold, sTs, yTs));
  computeMatUpdate();
  return PO_OK;
}

// ------------------------------------------------------------------------------------------------
int LSR1::updateWithZTs(Vec *s, Vec *y, const double *zTs, int *rc) {  // :636-747
  *rc = 0;
  const int mold = msub;
  std::vector<double> dots(2 * mold + 2, 0.0), d2(mold + 2, 0.0);
  const bool hint = zTs && (int)Z.size() == mold;
  BatchScope batch(ctx);  // both reductions (and whatever the caller has queued) share one collective + sync
  if (hint) {
    // Z_j = Y_j - b0 S_j with the b0 still in place: Y_j.s = Z_j.s + b0 S_j.s, so only the S columns are streamed
    std::vector<const double *> vp;
    for (int i = 0; i < mold; i++) vp.push_back(S[i]->d);
    vp.push_back(s->d);
    vp.push_back(y->d);
    PO_TRY(k_mdot(ctx, s->d, vp.data(), mold + 2, n, d2.data()));
  } else {
    std::vector<const double *> vp;
    for (int i = 0; i < mold; i++) vp.push_back(S[i]->d);
    for (int i = 0; i < mold; i++) vp.push_back(Y[i]->d);
    vp.push_back(s->d);
    vp.push_back(y->d);
    PO_TRY(k_mdot(ctx, s->d, vp.data(), 2 * mold + 2, n, dots.data()));
  }
  double yTy = 0.0;
  PO_TRY(k_reduce1(ctx, RED_SUMSQ, y->d, nullptr, n, &yTy));
  PO_TRY(batch.end());
  if (hint) {
    for (int i = 0; i < mold; i++) {
      dots[i] = d2[i];
      dots[mold + i] = zTs[i] + b0 * d2[i];
    }
    dots[2 * mold] = d2[mold];
    dots[2 * mold + 1] = d2[mold + 1];
  }
  const double sTs = dots[2 * mold], sTy = dots[2 * mold + 1];
  const double epsilon_precision = 1e-12;
  b0 = (sTy > epsilon_precision * yTy) ? yTy / sTy : 1.0;
  PO_TRY(storePair(s, y, dots.data(), dots.data() + mold, sTs, sTy));
  rebuildCompact();
  return PO_OK;
}

void LSR1::rebuildCompact() {  // M = b0 B - L - L^T - D (:712-728), Z_i = Y_i - b0 S_i, LU of M (:743)
  const int k = msub, m = msub_max;
  M.assign((size_t)k * k, 0.0);
  for (int i = 0; i < k; i++)
    for (int j = 0; j < k; j++) M[i + (size_t)k * j] += b0 * B[i + (size_t)m * j];
  for (int i = 0; i < k; i++) {
    for (int j = 0; j < i; j++) {
      M[i + (size_t)k * j] -= L[i + (size_t)m * j];
      M[j + (size_t)k * i] -= L[i + (size_t)m * j];
    }
  }
  for (int i = 0; i < k; i++) M[(size_t)i * (k + 1)] -= D[i];
  // Z_i = Y_i - b0 S_i for every held pair (:730-735): formed lazily -- by the interior point's
  // weighted-Gram pass when it is the next consumer, by ensureZ() otherwise
  Z.clear();
  d0.assign(k, 1.0);
  for (int i = 0; i < k; i++) Z.push_back(Zown[i]);
  z_pending = k > 0;
  factorM();
}

bool LSR1::pendingZ(std::vector<const double *> *Yp, std::vector<const double *> *Sp,
                    std::vector<double *> *Zout, double *b0_) const {
  if (!z_pending) return false;
  const int k = (int)Z.size();
  Yp->clear();
  Sp->clear();
  Zout->clear();
  for (int i = 0; i < k; i++) {
    Yp->push_back(Y[i]->d);
    Sp->push_back(S[i]->d);
    Zout->push_back(Zown[i]->d);
  }
  *b0_ = b0;
  return true;
}

int LSR1::ensureZ() const {
  if (!z_pending) return PO_OK;
  const int k = (int)Z.size();
  std::vector<double *> zd;
  std::vector<const double *> yp, sp;
  for (int i = 0; i < k; i++) {
    zd.push_back(Zown[i]->d);
    yp.push_back(Y[i]->d);
    sp.push_back(S[i]->d);
  }
  z_pending = false;
  return k_panel_lincomb(ctx, zd.data(), 1.0, yp.data(), -b0, sp.data(), k, n);
}

}  // namespace po
