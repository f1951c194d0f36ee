// Launchers for the sparse ("weighting") constraint blocks, nwblock = 1 (wcon.hip).
#pragma once
#include "core.hpp"

namespace po {

// the five w-sized blocks of ParOptVars (src/ParOptInteriorPoint.h:391-399), device pointers
struct WVars {
  double *zw, *sw, *tw, *zsw, *ztw;
};
struct PtrTableW {
  double *p[kMaxPanel];
};

// Constraint i (local) acts on the local variables start + i*(nw+skip) + [0, nw):
// cw_i = 1 - sum of those variables (examples/rosenbrock/rosenbrock.cpp:131-184).
struct GroupMap {
  int64_t nwcon = 0, start = 0;
  int nw = 0, skip = 0;
};

int k_group_sum(Ctx *c, const GroupMap &m, double *out, int init, double cst, double alpha,
                const double *v, int recip = 0);
// Cw = 1 / (Cdiag(v) + group sums of d): the factor of the scalar block form in one launch
int k_group_factor(Ctx *c, const GroupMap &m, const WVars &v, const double *d, double *cw);
// the structured K0^-1 apply in one launch (see wcon.hip); *done = false when the map does not tile (the caller then
// takes the three-launch form)
int k_group_k0(Ctx *c, const GroupMap &m, const double *d, const double *bx, const double *cw, const double *bw,
               double alpha, int64_t n, double *yx, double *yw, bool *done);
// out = -cw o (sum_j alpha_j U_j); acc += out when acc != nullptr
int k_w_correction(Ctx *c, const double *const *U, int nv, const double *alpha, const double *cw, int64_t w,
                   double *out, double *acc);
int k_group_scatter(Ctx *c, const GroupMap &m, double *out, double alpha, const double *w, int64_t n);
// the same with `out` overwritten (zero outside the groups): no separate fill pass
int k_group_scatter_set(Ctx *c, const GroupMap &m, double *out, double alpha, const double *w, int64_t n);
int k_group_panel(Ctx *c, const GroupMap &m, const double *const *P, int nv, const double *d,
                  double alpha, double *const *U);

int k_group_apply(Ctx *c, const GroupMap &m, const double *d, const double *bx, double alpha, const double *yw,
                  int64_t n, double *yx);
int k_w_apply_mid(Ctx *c, const double *cw, const double *bw, const double *u, int64_t w, double *yw);
// nwblock > 1: packed upper blocks of Cw (wcon.hip)
int k_blk_init(Ctx *c, const double *cdiag, int64_t nblocks, int B, double *blk);
int k_blk_factor(Ctx *c, double *blk, int64_t nblocks, int B, int *flag);
int k_blk_solve(Ctx *c, const double *blk, int64_t nblocks, int B, double *const *Y, int nv, int mode);
int k_mul(Ctx *c, double *y, double a, const double *x1, const double *x2, int64_t n);
int k_recip(Ctx *c, double *y, int64_t n);

// cw: the sparse constraint values (null: they sit in r.zw, which is overwritten)
// d2out != nullptr: also d2 of the block solve that follows (k_w_d2 of the blocks just written)
int k_w_res(Ctx *c, const WVars &v, const WVars &r, const double *gsw, const double *gtw, double mu,
            int64_t w, double out[12], const double *cw = nullptr, double *d2out = nullptr);
int k_w_scale5(Ctx *c, const WVars &dst, const WVars &src, double alpha, int64_t w);
int k_w_sumsq5(Ctx *c, const WVars &r, int64_t w, double out[5]);
int k_w_cdiag(Ctx *c, const WVars &v, int64_t w, double *cd);
int k_w_d2(Ctx *c, const WVars &v, const WVars &b, int64_t w, double *d2);
// out = {min_x, min_z}; comp != 0: {S10, S01, S11, min_x, min_z} (complementarity polynomial of the step, wcon.hip)
int k_w_step(Ctx *c, const WVars &v, const WVars &b, const double *dzw, int refine, double tau,
             const WVars &p, int64_t w, double *out, int comp = 0);
int k_w_res_step(Ctx *c, const WVars &v, const WVars &p, const WVars &r, int64_t w, double *d2out = nullptr);
int k_w_corrector(Ctx *c, const WVars &p, const WVars &r, int64_t w);
int k_w_comp_step(Ctx *c, const WVars &v, const WVars &p, double ax, double az, int64_t w, double *out);
int k_w_merit(Ctx *c, const WVars &v, const WVars &p, double sx, const double *gsw, const double *gtw,
              const double *cw, const double *awpx, int64_t w, double out[10]);
int k_w_trial(Ctx *c, const WVars &v, const WVars &p, double a, double eps, const double *gsw,
              const double *gtw, const double *cwt, int64_t w, double out[5]);
int k_w_update(Ctx *c, const WVars &v, const WVars &p, double ax, double az, double eps, int64_t w);
int k_w_affine(Ctx *c, const WVars &v, const WVars &p, double amin, int64_t w);
int k_w_clip(Ctx *c, double *zw, const double *src, const double *gsw, const double *gtw, int64_t w);
int k_w_gamma(Ctx *c, double *gsw, double *gtw, double gamma, int64_t nwineq, int64_t w);

}  // namespace po
