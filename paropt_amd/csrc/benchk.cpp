// po_bench_kernels: every hot kernel of one interior-point iteration timed in isolation on synthetic vectors
// (HIP events on the context's stream around each call; the calls include their reduction finish, ~20 us).
// Used by tools/microbench.py to tune kernels without running the whole solver.
#include <stdio.h>

#include <string>
#include <vector>

#include "core.hpp"
#include "qn.hpp"

using namespace po;

namespace {
struct Timer {
  Ctx *c;
  std::string out;
  int reps;
  template <class F>
  int run(const char *name, double bytes, double flops, F f) {
    double best = 1e30, sum = 0.0;
    for (int r = 0; r < reps + 1; r++) {
      PO_HIP(hipEventRecord(c->ev0, c->stream));
      PO_TRY(f());
      PO_HIP(hipEventRecord(c->ev1, c->stream));
      PO_HIP(hipEventSynchronize(c->ev1));
      float ms = 0.f;
      PO_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
      if (r == 0) continue;  // warm-up
      sum += ms;
      if (ms < best) best = ms;
    }
    char line[512];
    const double avg = sum / reps;
    snprintf(line, sizeof(line),
             "%s{\"kernel\": \"%s\", \"avg_ms\": %.4f, \"min_ms\": %.4f, \"alg_GB\": %.3f, \"GBps\": %.1f, "
             "\"frac_hbm_8TBps\": %.3f, \"TFLOPs\": %.2f}",
             out.empty() ? "" : ",\n ", name, avg, best, bytes * 1e-9, bytes / (avg * 1e-3) * 1e-9,
             bytes / (avg * 1e-3) * 1e-9 / 8000.0, flops / (avg * 1e-3) * 1e-12);
    out += line;
    return PO_OK;
  }
};
}  // namespace

extern "C" int po_bench_kernels(po_ctx ctx, int64_t n, int c, int k, int reps, char *report, int report_len) {
  if (!ctx || !report || n < 1 || c < 0 || k < 0 || c + k + 1 > kMaxPanel || k > 12 || reps < 1) return PO_ERR_ARG;
  Ctx *cx = ctx;
  std::vector<Vec *> all;
  auto mk = [&](uint64_t aid, double scale, double shift) -> Vec * {
    Vec *v = vec_new(cx, n);
    if (v) {
      all.push_back(v);
      (void)k_fill_hash(cx, v->d, n, 7, aid, 0, scale, shift);
    }
    return v;
  };
  Vec *x = mk(1, 0.9, 0.05), *lb = mk(2, 0.0, 0.0), *ub = mk(3, 0.0, 1.0), *zl = mk(4, 1.0, 0.5),
      *zu = mk(5, 1.0, 0.5), *g = mk(6, 2.0, -1.0), *rx = mk(7, 2.0, -1.0), *dinv = mk(8, 1.0, 0.5),
      *t = mk(9, 2.0, -1.0), *px = mk(10, 2.0, -1.0), *pzl = mk(11, 2.0, -1.0), *pzu = mk(12, 2.0, -1.0),
      *va = mk(13, 0.0, 0.0), *xt = mk(14, 0.0, 0.0), *yq = mk(15, 0.0, 0.0), *t2 = mk(16, 0.0, 0.0);
  std::vector<const double *> P, Yp, Sp;
  std::vector<double *> Zo;
  bool ok = x && lb && ub && zl && zu && g && rx && dinv && t && px && pzl && pzu && va && xt && yq && t2;
  for (int j = 0; ok && j < c + k; j++) {
    Vec *v = mk(100 + j, 2.0, -1.0);
    ok = v != nullptr;
    if (ok) P.push_back(v->d);
  }
  for (int j = 0; ok && j < k; j++) {
    Vec *y = mk(300 + j, 2.0, -1.0), *s = mk(400 + j, 2.0, -1.0);
    ok = y && s;
    if (ok) {
      Yp.push_back(y->d);
      Sp.push_back(s->d);
      Zo.push_back(const_cast<double *>(P[c + j]));
    }
  }
  int rc = ok ? PO_OK : PO_ERR_HIP;
  if (ok) {
    Bounds b;
    b.x = x->d; b.lb = lb->d; b.ub = ub->d; b.zl = zl->d; b.zu = zu->d;
    b.max_bound = 1e20; b.use_lower = 1; b.use_upper = 1;
    const int m = c + k;
    const double N = (double)n;
    std::vector<double> coef(kMaxPanel, 1e-3), coef2(kMaxPanel, 2e-3), out(kMaxPanel + 16, 0.0);
    std::vector<double> W((size_t)(m + 1) * (m + 1), 0.0);
    Timer T{cx, "", reps};
    auto seq = [&]() -> int {
      std::vector<const double *> Pt(P);
      Pt.push_back(t->d);
      PO_TRY(T.run("mdot(c)", 8.0 * (c + 1) * N, 2.0 * c * N,
                   [&] { return k_mdot(cx, x->d, P.data(), c, n, out.data()); }));
      PO_TRY(T.run("mdot(c+k)", 8.0 * (m + 1) * N, 2.0 * m * N,
                   [&] { return k_mdot(cx, t->d, P.data(), m, n, out.data()); }));
      PO_TRY(T.run("wgram(c+k)", 8.0 * (m + 1) * N, (double)m * (m + 1) * N,
                   [&] { return k_wgram(cx, dinv->d, P.data(), m, n, W.data()); }));
      PO_TRY(T.run("wgram(c+k+t)", 8.0 * (m + 2) * N, (double)(m + 1) * (m + 2) * N, [&] {
        return k_wgram(cx, dinv->d, Pt.data(), m + 1, n, W.data(), nullptr, nullptr, 0, 0.0, 1);
      }));
      if (k > 0) {
        // [Z | Ac | t] with the k L-SR1 columns formed in the pass (reads Y, S; writes Z)
        std::vector<const double *> P2(Yp);
        for (int j = 0; j < c; j++) P2.push_back(P[j]);
        P2.push_back(t->d);
        PO_TRY(T.run("wgram(c+k+t, k columns formed)", 8.0 * (m + 2 + 2 * k) * N, (double)(m + 1) * (m + 2) * N, [&] {
          return k_wgram(cx, dinv->d, P2.data(), m + 1, n, W.data(), Sp.data(), Zo.data(), k, 0.37, 1);
        }));
        for (int j = 0; j < k; j++) (void)k_fill_hash(cx, Zo[j], n, 7, 100 + c + j, 0, 2.0, -1.0);
      }
      if (c > 0) {  // the problem's Jacobian rewrite: c columns copied (scaled) in one launch
        std::vector<double *> dstp;
        for (int j = 0; j < c; j++) dstp.push_back(const_cast<double *>(P[j]));
        std::vector<const double *> srcp(P.begin(), P.begin() + c);
        // (source and destination panels must differ: use the first c/2 columns as sources of the second half)
        const int h = c / 2;
        if (h > 0)
          PO_TRY(T.run("panel_lincomb(c/2 columns)", 8.0 * 2 * h * N, 0.0, [&] {
            return k_panel_lincomb(cx, dstp.data() + h, -1.0, srcp.data(), 0.0, nullptr, h, n);
          }));
      }
      PO_TRY(T.run("kkt_res", 8.0 * (c + 8) * N, 0.0,
                   [&] { return k_kkt_res(cx, b, g->d, P.data(), coef.data(), c, 1e-3, n, t2->d, out.data()); }));
      PO_TRY(T.run("dinv", 8.0 * 6 * N, 0.0, [&] { return k_dinv(cx, b, 1.0, n, t2->d); }));
      PO_TRY(T.run("d1", 8.0 * 8 * N, 0.0, [&] { return k_d1(cx, b, rx->d, dinv->d, 1e-3, n, t2->d); }));
      PO_TRY(T.run("solve2_dots", 8.0 * (m + 13) * N, 0.0, [&] {
        return k_solve2_dots(cx, b, t->d, dinv->d, coef.data(), coef2.data(), P.data(), m, 1e-3, 0.95, rx->d, 1.0, n,
                             px->d, pzl->d, pzu->d, t2->d, va->d, c, out.data());
      }));
      PO_TRY(T.run("solve2_dots(step not stored)", 8.0 * (m + 9) * N, 0.0, [&] {
        return k_solve2_dots(cx, b, t->d, dinv->d, coef.data(), coef2.data(), P.data(), m, 1e-3, 0.95, rx->d, 1.0, n,
                             px->d, pzl->d, pzu->d, t2->d, va->d, c, out.data(), nullptr, 0);
      }));
      PO_TRY(T.run("solve2r(refine, first step recomputed)", 8.0 * (m + 12) * N, 0.0, [&] {
        return k_solve2r(cx, b, t->d, t2->d, dinv->d, coef.data(), coef2.data(), P.data(), m, 1e-3, 0.95, n, px->d,
                         pzl->d, pzu->d, va->d, c, out.data());
      }));
      PO_TRY(T.run("solve2_dots(nothing stored)", 8.0 * (m + 8) * N, 0.0, [&] {
        return k_solve2_dots(cx, b, t->d, dinv->d, coef.data(), coef2.data(), P.data(), m, 1e-3, 0.95, rx->d, 1.0, n,
                             px->d, pzl->d, pzu->d, nullptr, va->d, c, out.data(), nullptr, 0);
      }));
      PO_TRY(T.run("solve2r(refine, step and rhs recomputed)", 8.0 * (m + 12) * N, 0.0, [&] {
        return k_solve2r(cx, b, t->d, nullptr, dinv->d, coef.data(), coef2.data(), P.data(), m, 1e-3, 0.95, n, px->d,
                         pzl->d, pzu->d, va->d, c, out.data(), coef2.data(), rx->d, 1.0);
      }));
      PO_TRY(T.run("solve2r(refine, recomputed, + merit sums)", 8.0 * (m + 13) * N, 0.0, [&] {
        return k_solve2r(cx, b, t->d, nullptr, dinv->d, coef.data(), coef2.data(), P.data(), m, 1e-3, 0.95, n, px->d,
                         pzl->d, pzu->d, va->d, c, out.data(), coef2.data(), rx->d, 1.0, 0, nullptr, 0, 0.0, g->d,
                         out.data());
      }));
      PO_TRY(T.run("solve2(refine)", 8.0 * (m + 14) * N, 0.0, [&] {
        return k_solve2(cx, b, t->d, dinv->d, coef.data(), P.data(), m, 1e-3, 1, 0.95, n, px->d, pzl->d, pzu->d,
                        out.data(), nullptr, rx->d, 1.0, t2->d, va->d, c);
      }));
      PO_TRY(T.run("solve2(first)", 8.0 * (m + 11) * N, 0.0, [&] {
        return k_solve2(cx, b, t->d, dinv->d, coef.data(), P.data(), m, 1e-3, 0, 0.95, n, px->d, pzl->d, pzu->d,
                        out.data(), nullptr, rx->d, 1.0, t2->d, va->d, c);
      }));
      // round 6: the predictor-corrector's corrector right-hand side and solve, each as ONE pass (panels of <= 15
      // columns: the first min(m, 15) columns here), beside the kernels they replace (corrector + d1 + mdot: rows
      // above / below; solve2(first) + comp_merit)
      {
        const int mc = m < kCorrDotsMax ? m : kCorrDotsMax;
        if (mc > 0) {
          PO_TRY(T.run("corrector", 8.0 * 8 * N, 0.0,
                       [&] { return k_corrector(cx, b, px->d, pzl->d, pzu->d, n, t2->d, yq->d); }));
          PO_TRY(T.run("corr_d1_dots(min(c+k,15) columns)", 8.0 * (mc + 11) * N, 2.0 * mc * N, [&] {
            return k_corr_d1_dots(cx, b, px->d, pzl->d, pzu->d, rx->d, dinv->d, 1e-3, P.data(), mc, n, t2->d, out.data());
          }));
          PO_TRY(T.run("solve2c(min(c+k,15) columns)", 8.0 * (mc + 15) * N, 0.0, [&] {
            return k_solve2c(cx, b, t->d, dinv->d, coef.data(), P.data(), mc, 1e-3, 0.95, n, px->d, pzl->d, pzu->d, va->d,
                             c < mc ? c : mc, g->d, 1, out.data());
          }));
          (void)k_fill_hash(cx, px->d, n, 7, 10, 0, 2.0, -1.0);  // (the solve overwrote the step: restore the inputs)
          (void)k_fill_hash(cx, pzl->d, n, 7, 11, 0, 2.0, -1.0);
          (void)k_fill_hash(cx, pzu->d, n, 7, 12, 0, 2.0, -1.0);
        }
      }
      PO_TRY(T.run("kkt_res_update(+ Dinv, t of the next solve)", 8.0 * (c + 13) * N, 0.0, [&] {
        return k_kkt_res_update(cx, b, g->d, P.data(), coef.data(), c, 1e-3, n, t2->d, out.data(), nullptr, zl->d, pzl->d,
                                zu->d, pzu->d, 0.0, 1e-14, nullptr, 0.0, nullptr, 0.0, nullptr, nullptr, 0.0, -1.0,
                                nullptr, 0.0, yq->d, xt->d, 1.0, 1e-3);
      }));
      PO_TRY(T.run("comp_merit", 8.0 * 9 * N, 0.0, [&] {
        return k_comp_merit(cx, b, px->d, pzl->d, pzu->d, 0.5, 0.5, g->d, n, out.data());
      }));
      PO_TRY(T.run("trial", 8.0 * 5 * N, 0.0,
                   [&] { return k_trial(cx, b, px->d, 1e-3, 1e-14, n, xt->d, out.data()); }));
      PO_TRY(T.run("update_mult_yqn", 8.0 * 9 * N, 0.0, [&] {
        return k_update_mult_yqn(cx, zl->d, pzl->d, zu->d, pzu->d, 1e-6, 1e-14, 1, 1, rx->d, va->d, 1e-6, n, yq->d);
      }));
      return PO_OK;
    };
    rc = seq();
    snprintf(report, (size_t)report_len, "[%s]", T.out.c_str());
  }
  for (Vec *v : all) vec_decref(v);
  return rc;
}


// ------------------------------------------------------------------------------------------------------------------
// po_bench_vec_api (round 5): every ParOptVec operation north_star names (dot / norm / maxabs / l1norm / axpy / scale /
// copyValues / set / zeroEntries, src/ParOptVec.cpp:32-204), the multi-vector axpy behind ParOptQuasiNewton::mult
// (:414-416) and LBFGS / LSR1 mult + multAdd (src/ParOptQuasiNewton.cpp:390-459, 760-809) at size n, each beside the
// CEILING of its stream mix on the same GPU in the same process: a trivial kernel that moves the same number of
// input and output streams (16 B per lane, non-temporal, same persistent grid) and does nothing else.  Algorithmic
// bytes per SURVEY.md 8d: dot 16n, axpy 24n, scale / copy 16n, norm / maxabs / l1norm / set 8n, maxpy 8 (nv + 2) n,
// quasi-Newton mult 8 (2k + 3) n.
// ------------------------------------------------------------------------------------------------------------------
namespace {
typedef double bk_f64x2 __attribute__((ext_vector_type(2)));
struct MixPtrs {
  const double *in[48];
  double *out[2];
};
template <int NIN, int NOUT>
__global__ void __launch_bounds__(kBlock) trivial_mix_kernel(MixPtrs P, int64_t npairs, double *sink) {
  bk_f64x2 carry = {0.0, 0.0};
  for (int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x; q < npairs; q += (int64_t)gridDim.x * kBlock) {
    bk_f64x2 acc = {1.0, 2.0};
#pragma unroll
    for (int j = 0; j < NIN; j++) acc += __builtin_nontemporal_load(reinterpret_cast<const bk_f64x2 *>(P.in[j] + 2 * q));
#pragma unroll
    for (int j = 0; j < NOUT; j++) __builtin_nontemporal_store(acc, reinterpret_cast<bk_f64x2 *>(P.out[j] + 2 * q));
    carry += acc;
  }
  if (NOUT == 0 && carry.x + carry.y == 1.234567e300) sink[0] = carry.x;  // keeps the loads of a read-only mix alive
}
template <int NIN, int NOUT>
int launch_mix(Ctx *c, const MixPtrs &P, int64_t n, int bpc, double *sink) {
  const int64_t npairs = n >> 1;  // whole pairs only: the vectors' padding is not part of the measurement
  hipLaunchKernelGGL((trivial_mix_kernel<NIN, NOUT>), dim3(grid_for(c, n, bpc)), dim3(kBlock), 0, c->stream, P, npairs,
                     sink);
  PO_HIP(hipGetLastError());
  return PO_OK;
}
int run_mix(Ctx *c, int nin, int nout, const MixPtrs &P, int64_t n, double *sink) {
  const int bpc = nin > 4 ? kBpcPanel : kBpcStream;
#define PO_MIX(I, O) \
  if (nin == I && nout == O) return launch_mix<I, O>(c, P, n, bpc, sink)
  PO_MIX(1, 0);
  PO_MIX(2, 0);
  PO_MIX(0, 1);
  PO_MIX(1, 1);
  PO_MIX(2, 1);
  PO_MIX(11, 0);
  PO_MIX(11, 1);
  PO_MIX(12, 1);
  PO_MIX(41, 0);
  PO_MIX(41, 1);
  PO_MIX(42, 1);
#undef PO_MIX
  set_error("po_bench_vec_api: no trivial kernel for the mix %d in / %d out", nin, nout);
  return PO_ERR_ARG;
}

struct ApiTimer {
  Ctx *c;
  std::string out;
  int reps;
  // events around `reps` back-to-back calls (the calls of reductions include their final stage and the host's wait)
  template <class F>
  int time(F f, double *avg_ms) {
    PO_TRY(f());  // warm-up
    PO_HIP(hipStreamSynchronize(c->stream));
    PO_HIP(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; r++) PO_TRY(f());
    PO_HIP(hipEventRecord(c->ev1, c->stream));
    PO_HIP(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    PO_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = ms / reps;
    return PO_OK;
  }
  // The same calls inside ONE BatchScope: the first stages run back to back, the final stages and the host's wait
  // happen once at the end (outside the timed region) -- the rate the solver sees, where reductions are batched; the
  // synchronous form above includes the final-stage launch and the host round trip of EVERY call (11-15 us, a fifth
  // of a single-vector reduction at n = 50 M).
  template <class F>
  int time_batched(F f, double *avg_ms) {
    PO_HIP(hipStreamSynchronize(c->stream));
    BatchScope batch(c);
    PO_HIP(hipEventRecord(c->ev0, c->stream));
    for (int r = 0; r < reps; r++) PO_TRY(f());
    PO_HIP(hipEventRecord(c->ev1, c->stream));
    PO_TRY(batch.end());
    PO_HIP(hipEventSynchronize(c->ev1));
    float ms = 0.f;
    PO_HIP(hipEventElapsedTime(&ms, c->ev0, c->ev1));
    *avg_ms = ms / reps;
    return PO_OK;
  }
  // kernel_ms < 0: no separate kernel-only figure (the call has no host round trip)
  void row(const char *name, const char *ref, double bytes, double ms, const char *mix, double ceil_ms, double ceil_bytes,
           double kernel_ms = -1.0) {
    char line[1024];
    const double best = kernel_ms > 0.0 ? kernel_ms : ms;
    const double gbps = bytes / (best * 1e-3) * 1e-9, cg = ceil_bytes / (ceil_ms * 1e-3) * 1e-9;
    snprintf(line, sizeof(line),
             "%s{\"op\": \"%s\", \"reference\": \"%s\", \"alg_GB\": %.4f, \"call_ms\": %.4f, \"kernel_ms\": %.4f, "
             "\"GBps\": %.1f, \"frac_hbm_8TBps\": %.3f, \"mix\": \"%s\", \"ceiling_ms\": %.4f, \"ceiling_GBps\": %.1f, "
             "\"frac_of_ceiling\": %.3f, \"synchronous_call_GBps\": %.1f}",
             out.empty() ? "" : ",\n ", name, ref, bytes * 1e-9, ms, best, gbps, gbps / 8000.0, mix, ceil_ms, cg,
             ceil_ms / best, bytes / (ms * 1e-3) * 1e-9);
    out += line;
  }
};
}  // namespace

extern "C" int po_bench_vec_api(po_ctx ctx, int64_t n, int reps, char *report, int report_len) {
  if (!ctx || !report || n < 2 || reps < 1) return PO_ERR_ARG;
  Ctx *cx = ctx;
  std::vector<Vec *> all;
  auto mk = [&](uint64_t aid, double scale, double shift) -> Vec * {
    Vec *v = vec_new(cx, n);
    if (v) {
      all.push_back(v);
      (void)k_fill_hash(cx, v->d, n, 11, aid, 0, scale, shift);
    }
    return v;
  };
  int rc = PO_OK;
  double *sink = nullptr;
  LBFGS *bfgs = nullptr;
  LSR1 *sr1 = nullptr;
  auto body = [&]() -> int {
    PO_HIP(hipMalloc((void **)&sink, 64));
    Vec *x = mk(1, 2.0, -1.0), *y = mk(2, 2.0, -1.0);
    if (!x || !y) return PO_ERR_HIP;
    std::vector<const double *> V;
    for (int j = 0; j < 41; j++) {
      Vec *v = mk(100 + j, 2.0, -1.0);
      if (!v) return PO_ERR_HIP;
      V.push_back(v->d);
    }
    const double N = (double)n;
    ApiTimer T{cx, "", reps};
    std::vector<double> coef(48, 1e-3), o(64, 0.0);
    MixPtrs P;
    for (int j = 0; j < 48; j++) P.in[j] = V[j % 41];
    P.in[0] = x->d;
    P.out[0] = y->d;
    P.out[1] = y->d;
    auto ceiling = [&](int nin, int nout, double *ms) -> int {
      return T.time([&] { return run_mix(cx, nin, nout, P, n, sink); }, ms);
    };
    double ms = 0.0, c10 = 0.0, c20 = 0.0, c01 = 0.0, c11 = 0.0, c21 = 0.0;
    PO_TRY(ceiling(1, 0, &c10));
    PO_TRY(ceiling(2, 0, &c20));
    PO_TRY(ceiling(0, 1, &c01));
    PO_TRY(ceiling(1, 1, &c11));
    PO_TRY(ceiling(2, 1, &c21));
    std::vector<double> rr((size_t)reps + 1, 0.0);  // one landing slot per queued reduction
    double km = 0.0;
    int slot = 0;
    auto red = [&](int kind, const double *b) {
      return [&, kind, b]() -> int { return k_reduce1(cx, kind, x->d, b, n, &rr[(size_t)(slot++ % (reps + 1))]); };
    };
    PO_TRY(T.time(red(RED_DOT, y->d), &ms));
    PO_TRY(T.time_batched(red(RED_DOT, y->d), &km));
    T.row("dot", "src/ParOptVec.cpp:124-143", 16.0 * N, ms, "2 in / 0 out", c20, 16.0 * N, km);
    PO_TRY(T.time(red(RED_SUMSQ, nullptr), &ms));
    PO_TRY(T.time_batched(red(RED_SUMSQ, nullptr), &km));
    T.row("norm", "src/ParOptVec.cpp:63-80", 8.0 * N, ms, "1 in / 0 out", c10, 8.0 * N, km);
    PO_TRY(T.time(red(RED_AMAX, nullptr), &ms));
    PO_TRY(T.time_batched(red(RED_AMAX, nullptr), &km));
    T.row("maxabs", "src/ParOptVec.cpp:85-101", 8.0 * N, ms, "1 in / 0 out", c10, 8.0 * N, km);
    PO_TRY(T.time(red(RED_ASUM, nullptr), &ms));
    PO_TRY(T.time_batched(red(RED_ASUM, nullptr), &km));
    T.row("l1norm", "src/ParOptVec.cpp:106-119", 8.0 * N, ms, "1 in / 0 out", c10, 8.0 * N, km);
    PO_TRY(T.time([&] { return k_axpy(cx, y->d, 1e-9, x->d, n); }, &ms));
    T.row("axpy", "src/ParOptVec.cpp:189-204", 24.0 * N, ms, "2 in / 1 out", c21, 24.0 * N);
    PO_TRY(T.time([&] { return k_scale(cx, y->d, n, 1.0000001); }, &ms));
    T.row("scale", "src/ParOptVec.cpp:177-184", 16.0 * N, ms, "1 in / 1 out", c11, 16.0 * N);
    PO_TRY(T.time([&] { return k_copy(cx, y->d, x->d, n); }, &ms));
    T.row("copyValues", "src/ParOptVec.cpp:46-56", 16.0 * N, ms, "1 in / 1 out", c11, 16.0 * N);
    PO_TRY(T.time([&] { return k_fill(cx, y->d, n, 0.25); }, &ms));
    T.row("set", "src/ParOptVec.cpp:32-36", 8.0 * N, ms, "0 in / 1 out", c01, 8.0 * N);
    PO_TRY(T.time([&] { return k_fill(cx, y->d, n, 0.0); }, &ms));
    T.row("zeroEntries", "src/ParOptVec.cpp:41", 8.0 * N, ms, "0 in / 1 out", c01, 8.0 * N);
    // headline kernel and the multi-vector axpy of ParOptQuasiNewton::mult, k = 10 and 40
    for (int k : {10, 40}) {
      double cm0 = 0.0, cm1 = 0.0, cq = 0.0, m_mdot = 0.0, m_maxpy = 0.0;
      PO_TRY(ceiling(k + 1, 0, &cm0));
      PO_TRY(ceiling(k + 1, 1, &cm1));  // y <- b0 x + sum: x + k columns in, y out (mult)
      PO_TRY(ceiling(k + 2, 1, &cq));   // y <- y + ...: y read as well (maxpy / multAdd)
      char nm[64], mix[64];
      std::vector<double> oo((size_t)(reps + 1) * 64, 0.0);
      int ms_slot = 0;
      auto md = [&]() -> int { return k_mdot(cx, x->d, V.data(), k, n, &oo[(size_t)(ms_slot++ % (reps + 1)) * 64]); };
      double km_mdot = 0.0;
      PO_TRY(T.time(md, &m_mdot));
      PO_TRY(T.time_batched(md, &km_mdot));
      snprintf(nm, sizeof(nm), "mdot(nvecs=%d)", k);
      snprintf(mix, sizeof(mix), "%d in / 0 out", k + 1);
      T.row(nm, "src/ParOptVec.cpp:152-170", 8.0 * (k + 1) * N, m_mdot, mix, cm0, 8.0 * (k + 1) * N, km_mdot);
      PO_TRY(T.time([&] { return k_panel_axpy(cx, y->d, 0.0, nullptr, 1.0, coef.data(), V.data(), k, n); }, &m_maxpy));
      snprintf(nm, sizeof(nm), "maxpy(nvecs=%d)", k);
      snprintf(mix, sizeof(mix), "%d in / 1 out", k + 1);
      T.row(nm, "src/ParOptQuasiNewton.cpp:414-416", 8.0 * (k + 2) * N, m_maxpy, mix, cm1, 8.0 * (k + 2) * N);
    }
    // quasi-Newton products: L-BFGS with 20 pairs (k = 40 columns: config 2) and L-SR1 with 10 (config 3)
    Vec *s = mk(900, 2.0, -1.0), *yv = mk(901, 0.0, 0.0);
    if (!s || !yv) return PO_ERR_HIP;
    bfgs = new LBFGS(cx, n, 20);
    sr1 = new LSR1(cx, n, 10);
    for (int i = 0; i < 20; i++) {  // y = 2 s + 0.1 u: positive curvature, every pair accepted
      int code = 0;
      PO_TRY(k_fill_hash(cx, s->d, n, 11, 1000 + i, 0, 2.0, -1.0));
      PO_TRY(k_fill_hash(cx, yv->d, n, 11, 2000 + i, 0, 0.1, 0.0));
      PO_TRY(k_axpy(cx, yv->d, 2.0, s->d, n));
      PO_TRY(bfgs->update(s, yv, &code));
      if (code != 0) {
        set_error("po_bench_vec_api: synthetic pair %d was not accepted (code %d)", i, code);
        return PO_ERR_ARG;
      }
      if (i < 10) PO_TRY(sr1->update(s, yv, &code));
    }
    PO_TRY(sr1->ensureZ());
    struct QnCase {
      CompactQuasiNewton *qn;
      const char *name, *name_add, *ref, *ref_add;
      int k;
    } cases[2] = {{bfgs, "LBFGS::mult(k=40)", "LBFGS::multAdd(k=40)", "src/ParOptQuasiNewton.cpp:390-418",
                   "src/ParOptQuasiNewton.cpp:431-459", 40},
                  {sr1, "LSR1::mult(k=10)", "LSR1::multAdd(k=10)", "src/ParOptQuasiNewton.cpp:760-778",
                   "src/ParOptQuasiNewton.cpp:791-809", 10}};
    for (const QnCase &q : cases) {
      double cm0 = 0.0, cm1 = 0.0, cm2 = 0.0;
      PO_TRY(ceiling(q.k + 1, 0, &cm0));
      PO_TRY(ceiling(q.k + 1, 1, &cm1));
      PO_TRY(ceiling(q.k + 2, 1, &cm2));
      char mix[96];
      PO_TRY(T.time([&] { return q.qn->mult(x, y); }, &ms));
      snprintf(mix, sizeof(mix), "%d in / 0 out + %d in / 1 out", q.k + 1, q.k + 1);
      T.row(q.name, q.ref, 8.0 * (2 * q.k + 3) * N, ms, mix, cm0 + cm1, 8.0 * (2 * q.k + 3) * N);
      PO_TRY(T.time([&] { return q.qn->multAdd(1e-9, x, y); }, &ms));
      snprintf(mix, sizeof(mix), "%d in / 0 out + %d in / 1 out", q.k + 1, q.k + 2);
      T.row(q.name_add, q.ref_add, 8.0 * (2 * q.k + 4) * N, ms, mix, cm0 + cm2, 8.0 * (2 * q.k + 4) * N);
    }
    snprintf(report, (size_t)report_len, "[%s]", T.out.c_str());
    return PO_OK;
  };
  rc = body();
  (void)hipStreamSynchronize(cx->stream);
  delete bfgs;
  delete sr1;
  if (sink) (void)hipFree(sink);
  for (Vec *v : all) vec_decref(v);
  return rc;
}
