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
