// paropt_amd internal core: context, device vector, reduction plumbing, kernel launchers.
//
// Everything in this directory is the PRODUCT: gfx950-only HIP, no CPU fallback.  The host
// control flow (quasi-Newton bookkeeping, interior-point iteration) lives in qn.cpp / ip.cpp
// and only ever touches n-sized data through the launchers declared here.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <functional>
#include <string>
#include <vector>

#include "../../include/paropt_amd.h"

namespace po {

void set_error(const char *fmt, ...);
extern thread_local int g_last_code;

#define PO_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      po::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,     \
                    __LINE__);                                                           \
      return PO_ERR_HIP;                                                                 \
    }                                                                                    \
  } while (0)

#define PO_TRY(expr)            \
  do {                          \
    int _rc = (expr);           \
    if (_rc != PO_OK) return _rc; \
  } while (0)

constexpr int kBlock = 256;          // 4 wavefronts of 64
constexpr int kMaxPanel = 96;        // widest panel [Ac | Z] (+ a few extras) a kernel accepts
constexpr int kMaxRed = 8192;        // widest reduction payload (wgram: 15 blocks * 256 + slack)
constexpr int kWgramMaxVecs = 80;    // 5 block rows of 16

enum CommKind { COMM_SELF = 0, COMM_RCCL = 1, COMM_CALLBACK = 2 };

struct Ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int num_cu = 256;
  int max_blocks = 2048;  // persistent-grid cap for streaming kernels
  // communicator
  int rank = 0, size = 1;
  CommKind comm_kind = COMM_SELF;
  void *rccl_comm = nullptr;
  int rccl_allreduce = 1;  // pure-sum reductions through ncclAllReduce (else everything through the all-gather)
  long n_allreduce = 0, n_allgather = 0;
  po_allgather_fn cb_allgather = nullptr;
  void *cb_user = nullptr;
  // reduction plumbing
  // First-stage partials, [slot][block].  `d_partials` is the region of the launch being issued: the arena itself
  // outside a batch; inside one (round 4) every reduction gets its OWN region of the arena (ensure_partials advances
  // `partials_cursor`), so that the final stages of all queued reductions run as ONE launch at the flush instead of
  // one launch per reduction.
  double *d_partials = nullptr;
  double *partials_base = nullptr;  // the arena
  size_t partials_cap = 0;          // doubles
  size_t partials_cursor = 0;       // doubles handed out to queued reductions
  size_t partials_last = 0;         // size of the region handed out last
  double *d_red = nullptr;       // [kMaxRed] rank-local reduced values
  double *d_gather = nullptr;    // [size * kMaxRed]
  double *h_red = nullptr;       // pinned [size * kMaxRed]
  double *h_red_dev = nullptr;   // the same memory as the device sees it: without a device-side collective the final
                                 // reduction stage writes its results there (no device-to-host copy command)
  // ... and raises a sequence number behind them (pinned host memory; the last workgroup of the stage, found with a
  // device ticket): the host polls it instead of calling hipStreamSynchronize -- 11 instead of 15.7 us per
  // launch + wait + dependent launch (tools/sync_probe.hip).  red_seq counts the flagged launches, red_seq_seen what
  // the host has waited for; PAROPT_AMD_NO_FLAG_POLL=1 switches it off.
  unsigned long long *h_flag = nullptr, *h_flag_dev = nullptr;
  unsigned *d_ticket = nullptr;
  unsigned long long red_seq = 0, red_seq_seen = 0;
  int h_red_coherent = 0;                   // h_red / h_flag were allocated with the explicit coherence flags
  long n_flag_waits = 0, n_flag_timeouts = 0;  // polled completions, and those that fell back to the stream sync
  // development aid (PAROPT_AMD_HOST_TRACE=1, printed when the context is destroyed): host time between the return of
  // a synchronising reduction and the next kernel launch (what the GPU idles for beyond the round trip itself), the
  // time spent inside the launch calls, and the time spent waiting for results
  bool host_trace = false, host_gap_open = false;
  double host_t_sync = 0.0, host_gap_s = 0.0, host_launch_s = 0.0, host_wait_s = 0.0;
  long host_gap_n = 0, host_launch_n = 0, host_gap_n20 = 0, host_gap_long_n = 0;
  double host_gap_long_s = 0.0, host_t_launch_end = 0.0, host_inter_s = 0.0, host_inter_s30 = 0.0;
  long host_inter_n = 0, host_inter_n30 = 0;
  std::vector<double *> partials_overflow;  // regions handed out while the arena was full (freed after the flush)
  hipEvent_t ev0 = nullptr, ev1 = nullptr;  // immediate timings (recorded, synchronised and read in one call)
  // two n-sized scratch vectors for panels wider than one kernel's argument tables (kernels.hip: collapse_range);
  // allocated on first use
  double *wide_scratch[2] = {nullptr, nullptr};
  int64_t wide_n = 0;
  hipEvent_t ev_mdot0 = nullptr, ev_mdot1 = nullptr;  // k_mdot's timing only: may stay pending inside an open batch
  long n_reductions = 0;  // statistics: host-synchronising reductions issued
  // Batched reductions (BatchScope): while batch_depth > 0, reduce_finish only launches the rank-local final stage
  // into d_red[batch_cursor...] and queues the segment; ONE collective + copy + host sync serves every queued
  // segment when the batch is flushed.  Host code that depends on a queued result runs through after_reduce().
  struct PendingRed {
    int off, nsum, nmin, nmax;
    double *host_out;
    const double *part;  // its first-stage partials
    int nblocks;
  };
  int batch_enabled = 1;  // po_ctx_set_reduction_batching / PAROPT_AMD_NO_BATCH=1: every reduction syncs by itself
  int batch_depth = 0;
  int batch_cursor = 0;
  std::vector<PendingRed> batch_pend;
  std::vector<std::function<void()>> batch_after;
  long n_batched = 0;  // statistics: reductions that shared another one's host sync
  long n_launches = 0;
  // algorithmic HBM bytes of the n-sized launches so far (every operand stream counted once per launch, what
  // bench.py's iteration-level roofline divides by the step time); `alg_bytes_user` is the part issued between
  // InteriorPoint::userBegin/userEnd, i.e. by the problem's own callbacks (built-in problems and C-ABI vector calls)
  double alg_bytes = 0.0, alg_bytes_user = 0.0;
  int in_user = 0;
  // live timing of the headline kernel: mdot launches of exactly this many vectors are bracketed by ev0/ev1
  int time_mdot_nv = 0;
  double mdot_ms = 0.0;
  long mdot_count = 0;
  bool mdot_timing_pending = false;  // a timed launch is queued in the open batch (its events are read at the flush)
  // the same for the weighted-Gram launches (po_ctx_time_wgram): [0] plain, [1] with L-SR1 columns formed in the pass
  int time_wgram = 0;
  double wgram_ms[2] = {0.0, 0.0};
  long wgram_count[2] = {0, 0};
  int wgram_cols[2] = {0, 0};
  double wgram_bytes[2] = {0.0, 0.0};  // algorithmic bytes of the timed launches
};

struct Vec {
  Ctx *ctx;
  int64_t n;
  double *d;
  int ref;
  double *h;       // pinned host mirror (lazy)
  int h_live = 0;  // the mirror has been handed out (po_vec_get_array) and is the authoritative copy: every
                   // C-ABI operation on the vector uploads it first / downloads the result (po_vec_release_array)
};

// A small table of raw device pointers passed by value as a kernel argument.
struct PtrTable {
  const double *p[kMaxPanel];
};
struct CoefTable {
  double a[kMaxPanel];
};
// Unformed L-SR1 columns at the head of a panel: column j < count has the value P.p[j] - b0 * s[j] (P.p[j] = Y_j,
// s[j] = S_j), formed in registers by the consumer -- never written to HBM (a write costs about four reads).
constexpr int kMaxVirt = 12;
struct VirtCols {
  const double *s[kMaxVirt];
  int count;
  double b0;
};

// Reduction finish: combine first-stage partials ([slot][nblocks] in ctx->d_partials) into
// host_out[nsum+nmin+nmax]; slots are ordered sums, then mins, then maxs.  Collective.
// Inside a BatchScope the call returns before host_out is valid unless `now` is set (which flushes the batch).
int reduce_finish(Ctx *c, int nblocks, int nsum, int nmin, int nmax, double *host_out, bool now = false);
bool red_direct(const Ctx *c);  // the final stages write straight into the pinned host buffer (no RCCL collective)
// collective + host sync for everything queued; runs the after_reduce() work in order (keep_partials: an overflow
// region that is still waiting for its final stage and must survive the flush)
int batch_flush(Ctx *c, const double *keep_partials = nullptr, size_t keep_len = 0);
void batch_abort(Ctx *c);  // forget everything queued (error paths)
// Host work that reads the result of the preceding reduce_finish: immediately outside a batch, at the flush inside.
template <class F>
inline void after_reduce(Ctx *c, F &&f) {
  if (c->batch_depth > 0 && !c->batch_pend.empty()) {
    c->batch_after.emplace_back(std::forward<F>(f));
  } else {
    f();
  }
}
// Opens a batch on construction (when `on`); end() flushes whatever is queued -- also segments queued by an
// enclosing scope, which is always safe: a flush only makes results available earlier.  Leaving the scope
// without end() (an error is propagating) drops what is queued, enclosing scopes' segments included: nothing is
// written to result locations whose owners may already have returned.
struct BatchScope {
  Ctx *c;
  bool open;
  explicit BatchScope(Ctx *c_, bool on = true) : c(c_), open(on && c_->batch_enabled) {
    if (open) c->batch_depth++;
  }
  void begin() {  // for a scope constructed with on = false: open it now
    if (!open && c->batch_enabled) {
      open = true;
      c->batch_depth++;
    }
  }
  int end() {
    if (!open) return PO_OK;
    open = false;
    c->batch_depth--;
    return batch_flush(c);
  }
  // Like end(), but inside an enclosing scope the flush is left to that scope (for a callee whose results are all
  // delivered through after_reduce()).
  int end_nested() {
    if (!open) return PO_OK;
    open = false;
    c->batch_depth--;
    return c->batch_depth > 0 ? PO_OK : batch_flush(c);
  }
  // For a callee whose results are only post-processed (not branched on): inside an enclosing scope the flush is
  // left to that scope and `post` (host work on the reduced values) runs there; otherwise flush and run it now.
  // The result locations must stay valid until the enclosing scope ends.
  template <class F>
  int end_then(F &&post) {
    if (open) {
      open = false;
      c->batch_depth--;
      if (c->batch_depth > 0) {
        after_reduce(c, std::forward<F>(post));
        return PO_OK;
      }
      PO_TRY(batch_flush(c));
    }
    post();
    return PO_OK;
  }
  ~BatchScope() {
    if (open) {
      c->batch_depth--;
      batch_abort(c);
    }
  }
  BatchScope(const BatchScope &) = delete;
  BatchScope &operator=(const BatchScope &) = delete;
};
// Kernel-variant switches for A/B measurements inside ONE process (tools/ab_switch.py): the value set through
// po_debug_set_switch wins, else the environment variable, else the default.  Not part of the interface.
enum DbgSwitch { SW_WGRAM_RS = 0, SW_LINCOMB_2D = 1, SW_REDO_DT = 2, SW_LEAN_STEP = 3, SW_WGRAM_PRIO = 4,
                 SW_WGRAM_ABLATE = 5, SW_FUSED_MERIT = 6, SW_REDO_DT1 = 7, SW_BPC3 = 8, SW_BPC4 = 9, SW_LINCOMB_BPC = 10,
                 SW_PERTURB_W = 11, SW_GS_PRODUCER = 12, SW_S2D_TWO = 13, SW_MPC_FUSE = 14, SW_MPC_POLY = 15, SW_SPEC_DT = 16, SW_COUNT = 17 };  // one entry per use: an A/B run changes one thing
int dbg_switch(int id, const char *env, int dflt);
double host_now();  // seconds, monotonic
// around a kernel launch when Ctx::host_trace is on (see Ctx): begin closes the gap behind the last synchronisation
inline double host_trace_begin(Ctx *c);
inline void host_trace_end(Ctx *c, double t0);
void dbg_switch_set(int id, int value);  // value < 0: back to environment / default
int ensure_partials(Ctx *c, size_t doubles);
inline double host_trace_begin(Ctx *c) {
  if (!c->host_trace) return 0.0;
  const double t = host_now();
  if (!c->host_gap_open && c->host_t_launch_end > 0.0) {  // two launches with no synchronisation in between
    const double g = t - c->host_t_launch_end;
    if (g < 500e-6) {
      c->host_inter_s += g;
      c->host_inter_n++;
      if (g > 30e-6) {
        c->host_inter_n30++;
        c->host_inter_s30 += g;
      }
    }
  }
  if (c->host_gap_open) {
    const double g = t - c->host_t_sync;
    if (g < 500e-6) {  // (longer: between solves, around callbacks of the caller -- counted apart)
      c->host_gap_s += g;
      c->host_gap_n++;
      if (g > 20e-6) c->host_gap_n20++;
    } else {
      c->host_gap_long_s += g;
      c->host_gap_long_n++;
    }
    c->host_gap_open = false;
  }
  return t;
}
inline void host_trace_end(Ctx *c, double t0) {
  if (!c->host_trace) return;
  c->host_t_launch_end = host_now();
  c->host_launch_s += c->host_t_launch_end - t0;
  c->host_launch_n++;
}
// `streams` n-sized fp64 operand streams read or written by the launch being issued
inline void count_bytes(Ctx *c, double streams, int64_t n) {
  const double b = 8.0 * streams * (double)n;
  c->alg_bytes += b;
  if (c->in_user) c->alg_bytes_user += b;
}
constexpr int kBpcPanel = 2;   // workgroups per CU, panel-streaming kernels (solve passes, residual, panel axpy)
constexpr int kBpcStream = 4;  // workgroups per CU, few-stream kernels
int grid_for(Ctx *c, int64_t n);           // persistent grid, kBpcStream workgroups per CU
int grid_for(Ctx *c, int64_t n, int bpc);  // ... with an explicit workgroups-per-CU cap

// A w-sized vector standing in for an n-sized panel column (round 4): element i of the column is
// 0.0 + scale * w[g] when variable i is one of the first nw variables of group g = i / period (groups start at
// variable 0, g < nwcon), else 0.0 -- exactly what k_group_scatter_set (wcon.hpp; Problem::setSparseJacobianTranspose
// of the structured Jacobian) would write into an n-sized vector, formed in registers by the pass that would have
// read that vector: one launch, one n-sized write and one n-sized read less per use.  The passes treat it as the LAST
// column of their panel (same accumulation order as when it was a stored column).  n < 2^31.
struct GroupCol {
  const double *w = nullptr;  // nullptr: no such column
  double scale = 0.0;
  unsigned period = 1, nw = 0;
  long long nwcon = 0;
};

struct GroupCols2 {  // up to two grouped columns behind a panel, with their coefficients in two coefficient sets
  GroupCol g[2];
  double ca[2] = {0.0, 0.0}, cb[2] = {0.0, 0.0};
  int count = 0;
};

// ---- vector kernels (kernels.hip) -----------------------------------------------------------
int k_fill(Ctx *c, double *y, int64_t n, double alpha);
int k_fill_hash(Ctx *c, double *y, int64_t n, uint64_t seed, uint64_t aid, int64_t offset,
                double scale, double shift);
int k_copy(Ctx *c, double *y, const double *x, int64_t n);
int k_sign(Ctx *c, double *y, const double *x, int64_t n);  // y_i = x_i >= 0 ? 1 : -1
int k_scale(Ctx *c, double *y, int64_t n, double alpha);
int k_axpy(Ctx *c, double *y, double alpha, const double *x, int64_t n);
// y <- a*x + b*y + sum_j alpha[j]*V[j]   (x may be null when a == 0; b == 0 never reads y)
// ybase != nullptr: the b*y term reads that grouped column instead of y (y is only written)
int k_panel_axpy(Ctx *c, double *y, double a, const double *x, double b, const double *alpha,
                 const double *const *V, int nv, int64_t n, const GroupCol *ybase = nullptr);
// y1 <- a1*x1 + sum_j c1[j]*V[j] and y2 <- a2*x2 + sum_j c2[j]*V[j] in ONE pass over V (nv <= kMaxPanel)
int k_panel_axpy2(Ctx *c, double *y1, double a1, const double *x1, const double *c1, double *y2, double a2,
                  const double *x2, const double *c2, const double *const *V, int nv, int64_t n);
// dst_j <- a*X_j + b*Y_j, j < nv, one launch (Y may be null)
int k_panel_lincomb(Ctx *c, double *const *dst, double a, const double *const *X, double b,
                    const double *const *Y, int nv, int64_t n);
enum Red1 { RED_DOT = 0, RED_SUMSQ = 1, RED_ASUM = 2, RED_AMAX = 3 };
int k_reduce1(Ctx *c, int kind, const double *x, const double *y, int64_t n, double *out);
int k_mdot(Ctx *c, const double *x, const double *const *V, int nv, int64_t n, double *out);
// out = {smallest, largest} entry of x[0..n) (x need not be a padded vector: the tail element is read alone)
int k_minmax(Ctx *c, const double *x, int64_t n, double out[2]);
// launch-only variant for the roofline bench (no host sync); result stays in partials
int k_stream_launch(Ctx *c, int kind, double *x, double *y, int64_t n);  // bench: 0 = x.y, 1 = y <- x
int k_mdot_launch(Ctx *c, const double *x, const double *const *V, int nv, int64_t n, int *nblocks);
// W = V^T diag(d) V.  With kpend > 0 the first kpend (<= 12) columns are L-SR1 columns still to be
// formed: V[j] = Y_j, S[j] = S_j, and Z_j = Y_j - b0 S_j is used for the Gram AND written to Zout[j].
// preweighted_last != 0: the last column V[nv-1] = t already carries its weight (t = Dinv o d1), so that
// W[i][nv-1] = V_i . t for i < nv-1 -- the panel dots P^T t of the following bordered solve ride in the same
// pass over P (the entry W[nv-1][nv-1] is sum t^2, unused).
// may_defer: inside an open BatchScope the entries arrive in W at the flush (W must live until then; follow-up host
// work goes through after_reduce); otherwise, and always for panels processed in blocks, W is complete on return.
// groups != nullptr: the sparse constraints are the structured pattern (one constraint per group of nw consecutive
// variables, first group at variable 0) and the pass ALSO writes the panel image U_j = alpha * (group sums of
// d o V_j) for the first `ncols` columns -- what k_group_panel (wcon.hpp) computes in a pass of its own, same bits.
// *groups_done tells whether it did (panels / maps the fused kernel does not cover leave it false: the caller then
// runs k_group_panel).
struct GramGroups {
  int64_t nwcon = 0, start = 0;
  int nw = 0, skip = 0;
  double alpha = 1.0;
  int ncols = 0;
  double *const *U = nullptr;
};
int k_wgram(Ctx *c, const double *d, const double *const *V, int nv, int64_t n, double *W,
            const double *const *S = nullptr, double *const *Zout = nullptr, int kpend = 0,
            double b0 = 0.0, int preweighted_last = 0, bool may_defer = false,
            const GramGroups *groups = nullptr, bool *groups_done = nullptr);
int wgram_debug_stamps(double out[8]);  // tuning aid: PAROPT_AMD_WGRAM_ABLATE=16
int k_wgram_launch(Ctx *c, const double *d, const double *const *V, int nv, int64_t n,
                   int *nblocks, int *nslots, const double *const *S = nullptr,
                   double *const *Zout = nullptr, int kpend = 0, double b0 = 0.0,
                   int preweighted_last = 0, const GramGroups *groups = nullptr, bool *groups_done = nullptr);

// ---- interior-point kernels -------------------------------------------------------------------
struct Bounds {  // the per-element data every bound-aware kernel needs
  const double *x, *lb, *ub, *zl, *zu;
  double max_bound;  // max_bound_value (1e20)
  int use_lower, use_upper;
};
// rx = [L]zl - [U]zu - g + sum z_j A_j ; out = {comp product, active-bound count, l1 rx, l1 rzl,
// l1 rzu, l2^2 rx, l2^2 rzl, l2^2 rzu, max|rx|, max|rzl|, max|rzu|} with
// rzl = -((x-lb) zl - beta*mu), rzu = -((ub-x) zu - beta*mu).
// (computeKKTRes :1337-1446 + computeComp :2742-2820 + computeResNorm :1588-1723)
// yqn != nullptr: the same pass also completes the quasi-Newton gradient difference, yqn += [lo]zl - [up]zu - rx
// beta_mu2 >= 0: out has 13 entries, the last two are max|rzl|, max|rzu| for that second barrier term
// gcol != nullptr: one more column with coefficient gcoef behind the nc columns of A
int k_kkt_res(Ctx *c, const Bounds &b, const double *g, const double *const *A, const double *z,
              int nc, double beta_mu, int64_t n, double *rx, double *out, double *yqn = nullptr,
              double beta_mu2 = -1.0, const GroupCol *gcol = nullptr, double gcoef = 0.0);
// k_update_mult_yqn + k_kkt_res(..., yqn) in one pass (see kernels.hip): the bound multipliers take their step
// zl <- max(zl + a pzl, eps) here, y_qn gets both brackets, rx / out are those of the new point
int k_kkt_res_update(Ctx *c, const Bounds &b, const double *g, const double *const *A, const double *z, int nc,
                     double beta_mu, int64_t n, double *rx, double *out, double *yqn, double *zl,
                     const double *pzl, double *zu, const double *pzu, double a, double eps, const double *va,
                     double az, double *acz, double az_acz,
                     // lean step (pxs != nullptr): pzl / pzu are not read but formed from the design step pxs, the
                     // old point xold and the old multipliers with the barrier term of the step's solve
                     const double *pxs = nullptr, const double *xold = nullptr, double beta_mu_step = 0.0,
                     double beta_mu2 = -1.0,  // as k_kkt_res
                     const GroupCol *gcol = nullptr, double gcoef = 0.0,  // as k_kkt_res
                     // dinv_out != nullptr: also Dinv (diagonal spec_diag) and t = Dinv o d1 (spec_beta_mu) of the new
                     // point, as k_dinv_d1 would form them from what this pass has just stored
                     double *dinv_out = nullptr, double *t_out = nullptr, double spec_diag = 0.0,
                     double spec_beta_mu = 0.0);
// the mu-dependent part only (when the barrier parameter changes): out = {comp product,
// count, max|rzl|, max|rzu|}
int k_res_norms(Ctx *c, const Bounds &b, double beta_mu, int64_t n, double out[11]);
// Dinv = 1/(diag + [L] zl/(x-lb) + [U] zu/(ub-x))   (setUpKKTDiagSystem :1864-1910)
int k_dinv(Ctx *c, const Bounds &b, double diag, int64_t n, double *dinv,
           const double *hdiag = nullptr);  // hdiag: per-element Hessian diagonal added to `diag`
// Newton-Krylov support: the alpha-scaled bordered solve (solveKKTDiagSystem :2441-2614)
int k_d1s(Ctx *c, const Bounds &b, const double *bx, const double *dinv, double alpha, double beta_mu,
          int64_t n, double *t);
int k_solve2s(Ctx *c, const Bounds &b, const double *t, const double *dinv, const double *coef,
              const double *const *P, int nv, double alpha, double beta_mu, int full, double tau, int64_t n,
              double *px, double *pzl, double *pzu, double out[2]);
// t = Dinv*(rx + [L] rzl/(x-lb) - [U] rzu/(ub-x)) with rzl, rzu recomputed from beta_mu
// (the d1 build of solveKKTDiagSystem :2091-2108 followed by mat->apply :2139)
int k_d1(Ctx *c, const Bounds &b, const double *rx, const double *dinv, double beta_mu, int64_t n,
         double *t, const double *cl = nullptr, const double *cu = nullptr);
// k_dinv followed by k_d1 (no corrector terms) in one pass over the bound data
int k_dinv_d1(Ctx *c, const Bounds &b, double diag, const double *hdiag, const double *rx, double beta_mu,
              int64_t n, double *dinv, double *t, int raw = 0);  // raw: t = d1 (not weighted by Dinv)
// Mehrotra corrector products of the affine step (addMehrotraCorrectorResidual :1765-1788)
int k_corrector(Ctx *c, const Bounds &b, const double *px, const double *pzl, const double *pzu,
                int64_t n, double *cl, double *cu);
// predictor-corrector strategy, corrector right-hand side in ONE pass (round 6): k_corrector + k_d1 + k_mdot with the
// bits of that sequence; (cl, cu) are not stored (k_solve2c re-forms them).  out = P^T t [nv], nv <= kCorrDotsMax
constexpr int kCorrDotsMax = 15;
int k_corr_d1_dots(Ctx *c, const Bounds &b, const double *px, const double *pzl, const double *pzu, const double *rx,
                   const double *dinv, double beta_mu, const double *const *V, int nv, int64_t n, double *t,
                   double *out);
// ... and the corrector solve that follows: k_solve2 (first solve, corrector terms re-formed from the affine step in
// (px, pzl, pzu), which it overwrites) + the sums of k_comp_merit in polynomial form.
// out[12] = {S10, S01, S11, ppos, pneg, g.px, px.px, pos log, neg log | max_x, max_z | max|px|}
int k_solve2c(Ctx *c, const Bounds &b, const double *t, const double *dinv, const double *alpha,
              const double *const *P, int nv, double beta_mu, double tau, int64_t n, double *px, double *pzl,
              double *pzu, double *va, int nca, const double *g, int want_logs, double out[12]);
// Second half of the bordered solve.  acc = sum_j alpha_j P_j ; dx = t + Dinv*acc.
//   first solve  (refine == 0): px = dx; pzl = [L](rzl - zl dx)/(x-lb); pzu = [U](rzu + zu dx)/(ub-x)
//   refinement   (refine == 1): r'zl = rzl - [L]((x-lb) pzl + px zl), r'zu likewise;
//                               px += dx; pzl += [L](r'zl - zl dx)/(x-lb); pzu += ...
// out = {max_x, max_z}: the fraction-to-boundary minima of computeMaxStep :2942-3103 for the
// final (px, pzl, pzu) with fraction tau (NOT masked by the bound predicates for px, as in the
// reference).
// When coef2 != nullptr (first pass only) the same panel pass also evaluates the refinement
// residual with coefficient set coef2 and writes t' = Dinv*d1' (see k_res_step) to tout (which may
// alias t): one panel pass less per iteration.
int k_solve2(Ctx *c, const Bounds &b, const double *t, const double *dinv, const double *alpha,
             const double *const *P, int nv, double beta_mu, int refine, double tau, int64_t n,
             double *px, double *pzl, double *pzu, double out[2], const double *coef2 = nullptr,
             const double *rx = nullptr, double diag = 0.0, double *tout = nullptr,
             double *va = nullptr, int nca = 0, const double *cl = nullptr,
             const double *cu = nullptr);
// first solve pass + refinement residual + its panel dots in ONE pass: out = {P^T t' [nv], max_x, max_z}.
// traw != nullptr: the RAW right-hand side d1' is stored there instead of t' = Dinv o d1' in tout (the sparse-constraint
// path applies its block solve to d1'); the dots are those of t' either way.
int k_solve2_dots(Ctx *c, const Bounds &b, const double *t, const double *dinv, const double *alpha,
                  const double *coef2, const double *const *P, int nv, double beta_mu, double tau,
                  const double *rx, double diag, int64_t n, double *px, double *pzl, double *pzu,
                  double *tout, double *va, int nca, double *out, double *traw = nullptr,
                  int store_step = 1,  // 0: nothing of the step, 1: px, pzl, pzu (and va), 2: px only
                  int ca0 = 0,  // the nca constraint columns are P[ca0 .. ca0 + nca)
                  const double *const *vs = nullptr, int nvirt = 0, double b0v = 0.0,  // P[j] - b0v vs[j], j < nvirt
                  double dinv_diag = 0.0,  // t == nullptr: Dinv (this diagonal) and t re-formed from the bound data and rx
                  // grouped columns behind P (panel positions nv, nv + 1): they enter the two row sums with (ca, cb);
                  // no panel dots are taken for them (out keeps its layout {dots[nv], max_x, max_z})
                  const GroupCols2 *gcols = nullptr);
// store_step == 0 above leaves (px, pzl, pzu, va) unwritten; this refinement pass recomputes that first step from
// (t1, a1) and applies the refinement (t2, a2) on top in ONE sweep over P: out = {max_x, max_z} of the final step
int k_solve2r(Ctx *c, const Bounds &b, const double *t1, const double *t2, const double *dinv, const double *a1,
              const double *a2, const double *const *P, int nv, double beta_mu, double tau, int64_t n, double *px,
              double *pzl, double *pzu, double *va, int nca, double out[2], const double *ar = nullptr,
              const double *rx = nullptr, double diag = 0.0,  // t2 == nullptr: t2 recomputed from (ar, rx, diag)
              int ca0 = 0,  // the nca constraint columns are P[ca0 .. ca0 + nca)
              const double *const *vs = nullptr, int nvirt = 0, double b0v = 0.0,  // P[j] - b0v vs[j], j < nvirt
              // g != nullptr (recomputed right-hand side form only): merit_out[10] = {S10, S01, S11, ppos, pneg, g.px,
              // px.px | max_x, max_z | max|px|} of the final step (see solve2r_kernel) instead of `out`
              const double *g = nullptr, double *merit_out = nullptr,
              // t1 == nullptr: Dinv (diagonal dinv_diag) and t1 re-formed from the bound data and rx in registers
              double dinv_diag = 0.0,
              // stored right-hand side form: one more column behind P with coefficients gc1 / gc2 in the two sets
              const GroupCol *gcol = nullptr, double gc1 = 0.0, double gc2 = 0.0);
// (pzl, pzu) of a lean step as vectors: [L] (rzl - zl px) / (x - lb), [U] (rzu + zu px) / (ub - x)
int k_form_pz(Ctx *c, const Bounds &b, const double *px, double beta_mu, int64_t n, double *pzl, double *pzu);
// multiplier update fused with y_qn = rx - [lo]zl_old + [up]zu_old + az*va (see kernels.hip)
int k_update_mult_yqn(Ctx *c, double *zl, const double *pzl, double *zu, const double *pzu, double a,
                      double eps, int use_lower, int use_upper, const double *rx, const double *va,
                      double az, int64_t n, double *yqn, double *acz = nullptr);
// Residual of the linearised KKT system for iterative refinement, already folded into the next
// solve's right-hand side:  r'x = rx - diag*px + sum coef_j P_j + [L]pzl - [U]pzu ;
// r'zl, r'zu as above ; t' = Dinv*(r'x + [L] r'zl/(x-lb) - [U] r'zu/(ub-x)).
// (computeKKTRes + addKKTResStep :1451-1583 + the d1 build of the following solve)
int k_res_step(Ctx *c, const Bounds &b, const double *rx, const double *px, const double *pzl,
               const double *pzu, const double *dinv, const double *coef, const double *const *P,
               int nv, double diag, double beta_mu, int64_t n, double *tprime);
// checkKKTStep :6212-6360: out = {max|r'x|, max|r'zl|, max|r'zu|} of the linearised KKT residual at the step
int k_step_check(Ctx *c, const Bounds &b, const double *rx, const double *px, const double *pzl, const double *pzu,
                 const double *coef, const double *const *P, int nv, double diag, double beta_mu, int64_t n,
                 double out[3]);
// computeCompStep :2825-2923 at (x + ax*px, zl + az*pzl, zu + az*pzu): out = {product, count}
int k_comp_step(Ctx *c, const Bounds &b, const double *px, const double *pzl, const double *pzu,
                double ax, double az, int64_t n, double out[2]);
// comp_step + merit0 (unscaled step) + max|px| in one pass: out = {comp product, count, pos log, neg log,
// ppos, pneg, g.px, px.px, max|px|}
int k_comp_merit(Ctx *c, const Bounds &b, const double *px, const double *pzl, const double *pzu, double ax,
                 double az, const double *g, int64_t n, double out[9]);
// evalMeritInitDeriv :3652-3714 barrier part + the three design-space inner products:
// out = {pos log, neg log, pos presult, neg presult, g.px, px.px}; px is scaled by sx.
int k_merit0(Ctx *c, const Bounds &b, const double *px, double sx, const double *g, int64_t n,
             double out[6]);
// trial point of the line search (:3997-4001) xt = clamp(x + a*px, lb+eps, ub-eps) and the
// log-barrier partial sums of evalMeritFunc :3541-3565 at xt: out = {pos, neg}
// sout != nullptr: also sout = a * px, the quasi-Newton step of this trial
int k_trial(Ctx *c, const Bounds &b, const double *px, double a, double eps, int64_t n, double *xt,
            double out[2], double *sout = nullptr);
// zl <- max(zl + a*pzl, eps) ; zu likewise  (computeStepAndUpdate :4186-4191)
int k_update_mult(Ctx *c, double *zl, const double *pzl, double *zu, const double *pzu, double a,
                  double eps, int use_lower, int use_upper, int64_t n);
// initAffineStepMultipliers :5630-5652: zl = [L] max(amin, |zl + pzl|) (unchanged otherwise)
int k_affine_mult(Ctx *c, const Bounds &b, double *zl, const double *pzl, double *zu,
                  const double *pzu, double amin, int64_t n);
// initAndCheckDesignAndBounds :4290-4360; out flag bits 1/2/4; zl/zu zeroed on inactive bounds
int k_check_bounds(Ctx *c, double *x, double *lb, double *ub, double *zl, double *zu,
                   double max_bound, double rel_bound, int both, int64_t n, int *flag);
int k_bounds_mode(Ctx *c, double *x, double *lb, double *ub, int mode, int64_t offset, int64_t n);
// entries at their clamp values: out = {#x == lb + eps, #x == ub - eps, #zl == eps, #zu == eps} (global sums)
int k_clamp_count(Ctx *c, const double *x, const double *lb, const double *ub, const double *zl, const double *zu,
                  double eps, int64_t n, double out[4]);
int k_zero_inactive(Ctx *c, const double *lb, const double *ub, double *zl, double *zu,
                    double max_bound, int64_t n);

// ---- built-in problems --------------------------------------------------------------------------
// f-parts: quadratic sum(0.5 q x^2 + b x), convex sum(b^2/(eps+x)), rosenbrock chain
int k_quadratic_f(Ctx *c, const double *q, const double *b, const double *x, int64_t n, double *f);
int k_quadratic_g(Ctx *c, const double *q, const double *b, const double *x, int64_t n, double *g);
int k_convex_f(Ctx *c, const double *b, const double *x, int64_t n, double *f);
int k_convex_g(Ctx *c, const double *b, const double *x, int64_t n, double *g);
int k_rosen_f(Ctx *c, const double *x, int64_t n, double out[3]);
int k_rosen_g(Ctx *c, const double *x, int64_t n, double *g, double *a0, double *a1);
// Lagrangian Hessians of the built-in problems: h = diag(H) (px == null) or H px
int k_sep_hess(Ctx *c, int kind, const double *q, const double *b, const double *x, const double *px,
               int64_t n, double *h);
int k_rosen_hess(Ctx *c, const double *x, double z0, const double *px, int64_t n, double *h);

// ---- host dense algebra (lu.cpp) --------------------------------------------------------------
// LAPACK-dgetf2-style LU with partial pivoting, column-major, pivots 0-based.  Returns info
// (index+1 of the first exactly-zero pivot, or 0); like the reference, callers ignore it
// (src/ParOptInteriorPoint.cpp:1968-1969).
int lu_factor(int n, double *A, int lda, int *piv);
void lu_solve(int n, const double *A, int lda, const int *piv, double *b);

}  // namespace po

struct po_ctx_s : po::Ctx {};
struct po_vec_s : po::Vec {};
