"""
TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's trust-region driver and its
subproblems, in the reference's operation order:

  * ParOptQuadraticSubproblem      src/ParOptTrustRegion.cpp:27-466
  * ParOptInfeasSubproblem         src/ParOptTrustRegion.cpp:468-650
  * ParOptTrustRegion (SL1QP with the adaptive penalty update; penalty_method strategy)
                                   src/ParOptTrustRegion.cpp:652-1687, 2391-2472
  * ParOptCompactEigenApprox, ParOptEigenQuasiNewton, ParOptEigenSubproblem
                                   src/ParOptCompactEigenvalueApprox.cpp:23-724

Pinned against tests/golden/tr_*.npz (trajectories of the compiled reference, oracle/ref_driver.cpp
mode "tr").  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
The filter globalisation (filterOptimize :1690-2210) is restated including its quirks; the
second-order correction is dead code in the reference (its only call site is commented out).
"""
import math

import numpy as np

from . import paropt_oracle as po

TR_DEFAULTS = dict(  # ParOptTrustRegion::addDefaultOptions :739-847
    tr_init_size=0.1,
    tr_min_size=1e-3,
    tr_max_size=1.0,
    tr_eta=0.25,
    tr_bound_relax=1e-4,
    tr_adaptive_gamma_update=1,
    tr_accept_step_strategy="penalty_method",
    tr_max_iterations=200,
    tr_l1_tol=1e-6,
    tr_linfty_tol=1e-6,
    tr_infeas_tol=1e-5,
    tr_penalty_gamma_max=1e4,
    tr_penalty_gamma_min=0.0,
    filter_sufficient_reduction=1,
    filter_gamma=1e-5,
    filter_has_feas_restore_phase=1,
    tr_use_soc=0,
    tr_adaptive_objective="linear_objective",
    tr_adaptive_constraint="linear_constraint",
    tr_steering_barrier_strategy="mehrotra_predictor_corrector",
    tr_steering_starting_point_strategy="affine_step",
    function_precision=1e-10,
    penalty_gamma=1000.0,
)


class _SubproblemBase:
    """Shared model storage of the two trust-region subproblems."""

    def __init__(self, prob):
        self.prob = prob
        self.comm = prob.comm
        self.nlocal, self.c = prob.nlocal, prob.c
        self.nwcon = getattr(prob, "nwcon", 0)
        self.nwineq = getattr(prob, "nwineq", 0)
        n = self.nlocal
        self.ops = po.VecOps(self.comm)
        self.xk = np.full(n, 0.5)
        self.lk, self.uk = np.zeros(n), np.ones(n)
        self.lb, self.ub = np.zeros(n), np.ones(n)
        self.fk = 0.0
        self.gk = np.zeros(n)
        self.ck = np.zeros(self.c)
        self.Ak = [np.zeros(n) for _ in range(self.c)]
        self.ft = 0.0
        self.gt = np.zeros(n)
        self.ct = np.zeros(self.c)
        self.At = [np.zeros(n) for _ in range(self.c)]
        self.qn_update_type = 0

    def set_trust_region_bounds(self, tr_size):  # :156-173
        self.lk = np.maximum(-tr_size, self.lb - self.xk)
        self.uk = np.minimum(tr_size, self.ub - self.xk)

    def vars_and_bounds(self):  # :278-285
        step = np.zeros(self.nlocal)
        step += 0.5 * self.lk
        step += 0.5 * self.uk
        return step, self.lk.copy(), self.uk.copy()

    # sparse constraints are linearised about xk (:345-385)
    def eval_sparse_con(self, step):
        out = self.prob.eval_sparse_con(self.xk)
        return self.prob.add_sparse_jacobian(1.0, step, out)

    def add_sparse_jacobian(self, alpha, px, out):
        return self.prob.add_sparse_jacobian(alpha, px, out)

    def add_sparse_jacobian_transpose(self, alpha, pzw, out):
        return self.prob.add_sparse_jacobian_transpose(alpha, pzw, out)

    def add_sparse_inner_product(self, alpha, cvec, A):
        return self.prob.add_sparse_inner_product(alpha, cvec, A)

    # createQuasiDefMat() forwards to the wrapped problem (src/ParOptTrustRegion.cpp:257-258,513-514): a CSR-form
    # problem keeps its general sparse quasi-definite matrix under the subproblem
    @property
    def csr_form(self):
        return bool(getattr(self.prob, "chain", None)) or bool(getattr(self.prob, "csr_form", False))

    def sparse_jacobian_dense(self):
        return self.prob.sparse_jacobian_dense()

    def _lagrangian_gradient_difference(self, z, zw):
        """t = [gt - At^T z - Aw^T zw] - [gk - Ak^T z - Aw^T zw]  (:187-205 / eigen :492-510)."""
        t = self.gt.copy()
        for i in range(self.c):
            t += -z[i] * self.At[i]
        if self.nwcon > 0:
            self.prob.add_sparse_jacobian_transpose(-1.0, zw, t)
        t += -1.0 * self.gk
        for i in range(self.c):
            t += z[i] * self.Ak[i]
        if self.nwcon > 0:
            self.prob.add_sparse_jacobian_transpose(1.0, zw, t)
        return t

    def reject_trial_step(self):  # :226-231
        self.ft = 0.0
        self.ct = np.zeros(self.c)


class QuadraticSubproblem(_SubproblemBase):
    def __init__(self, prob, qn):
        super().__init__(prob)
        self.qn = qn

    def get_quasi_newton(self):
        return self.qn

    def init_model_and_bounds(self, tr_size):  # :141-151
        x, lb, ub = self.prob.vars_and_bounds()
        self.xk, self.lb, self.ub = x.copy(), lb.copy(), ub.copy()
        self.set_trust_region_bounds(tr_size)
        _, self.fk, ck = self.prob.eval_obj_con(self.xk)
        self.ck = np.array(ck, dtype=float)
        _, self.gk, self.Ak = self.prob.eval_obj_con_gradient(self.xk)

    def eval_trial_step_and_update(self, update_flag, step, z, zw):  # :175-212
        xtemp = self.xk.copy()
        xtemp += 1.0 * step
        fail, self.ft, ct = self.prob.eval_obj_con(xtemp)
        self.ct = np.array(ct, dtype=float)
        _, self.gt, self.At = self.prob.eval_obj_con_gradient(xtemp)
        if self.qn is not None and update_flag:
            t = self._lagrangian_gradient_difference(z, zw)
            self.qn_update_type = self.qn.update(step.copy(), t)
        return fail, self.ft, self.ct.copy()

    def accept_trial_step(self, step, z, zw):  # :214-224
        self.fk = self.ft
        self.xk = self.xk + 1.0 * step
        self.gk = self.gt.copy()
        self.ck = self.ct.copy()
        self.Ak = [a.copy() for a in self.At]

    def eval_obj_con(self, step):  # :290-323
        if step is None:
            return 0, self.fk, self.ck.copy()
        fobj = self.fk + self.ops.dot(self.gk, step)
        if self.qn is not None:
            t = self.qn.mult(step)
            fobj += 0.5 * self.ops.dot(step, t)
        cons = np.array([self.ck[i] + self.ops.dot(self.Ak[i], step) for i in range(self.c)])
        return 0, fobj, cons

    def eval_obj_con_gradient(self, step):  # :328-343
        Ac = [a.copy() for a in self.Ak]
        if self.qn is not None:
            g = self.qn.mult(step)
            g = g + 1.0 * self.gk
        else:
            g = self.gk.copy()
        return 0, g, Ac


class InfeasSubproblem:
    """Steering problem of the adaptive penalty update (:468-650)."""

    def __init__(self, sub, objective, constraint):
        self.sub = sub
        self.comm = sub.comm
        self.nlocal, self.c = sub.nlocal, sub.c
        self.nwcon, self.nwineq = sub.nwcon, sub.nwineq
        self.objective, self.constraint = objective, constraint
        self.obj_scale = 1.0
        self.ops = sub.ops

    def vars_and_bounds(self):
        return self.sub.vars_and_bounds()

    def eval_obj_con(self, step):  # :541-580
        s = self.sub
        if self.objective == "subproblem_objective" or self.constraint == "subproblem_constraint":
            _, fobj, cons = s.eval_obj_con(step)
        else:
            fobj, cons = 0.0, np.zeros(self.c)
        if self.objective == "linear_objective":
            fobj = s.fk + self.ops.dot(s.gk, step)
        elif self.objective == "constant_objective":
            fobj = s.fk
        if self.constraint == "linear_constraint":
            cons = np.array([s.ck[i] + self.ops.dot(s.Ak[i], step) for i in range(self.c)])
        return 0, fobj * self.obj_scale, cons

    def eval_obj_con_gradient(self, step):  # :585-612
        s = self.sub
        g, Ac = None, None
        if self.objective == "subproblem_objective" or self.constraint == "subproblem_constraint":
            _, g, Ac = s.eval_obj_con_gradient(step)
        if self.objective == "linear_objective":
            g = s.gk.copy()
        elif self.objective == "constant_objective":
            g = np.zeros(self.nlocal)
        if self.constraint == "linear_constraint":
            Ac = [a.copy() for a in s.Ak]
        return 0, g * self.obj_scale, Ac

    def eval_sparse_con(self, step):
        return self.sub.eval_sparse_con(step)

    def add_sparse_jacobian(self, alpha, px, out):
        return self.sub.add_sparse_jacobian(alpha, px, out)

    def add_sparse_jacobian_transpose(self, alpha, pzw, out):
        return self.sub.add_sparse_jacobian_transpose(alpha, pzw, out)

    def add_sparse_inner_product(self, alpha, cvec, A):
        return self.sub.add_sparse_inner_product(alpha, cvec, A)

    # createQuasiDefMat() forwards to the wrapped problem (src/ParOptTrustRegion.cpp:257-258,513-514): a CSR-form
    # problem keeps its general sparse quasi-definite matrix under the subproblem
    @property
    def csr_form(self):
        return bool(getattr(self.sub, "chain", None)) or bool(getattr(self.sub, "csr_form", False))

    def sparse_jacobian_dense(self):
        return self.sub.sparse_jacobian_dense()


# ---- compact eigenvalue approximation -------------------------------------------------------------
class CompactEigenApprox:  # :23-120
    def __init__(self, n, N, ops):
        self.N, self.ops = N, ops
        self.c0 = 0.0
        self.g0 = np.zeros(n)
        self.M = np.zeros((N, N))
        self.Minv = np.zeros((N, N))
        self.hvecs = [np.zeros(n) for _ in range(N)]

    def mult_add(self, alpha, x, y):  # :52-64 (y updated in place)
        tmp = self.ops.mdot(x, self.hvecs)
        for i in range(self.N):
            scale = 0.0
            for j in range(self.N):
                scale += self.M[i, j] * tmp[j]
            y += (alpha * scale) * self.hvecs[i]
        return y

    def eval_approximation(self, s):  # :92-106
        c = self.c0
        if s is not None:
            c += self.ops.dot(self.g0, s)
            tmp = self.ops.mdot(s, self.hvecs)
            for i in range(self.N):
                for j in range(self.N):
                    c += 0.5 * self.M[i, j] * tmp[i] * tmp[j]
        return c

    def eval_approximation_gradient(self, s):  # :108-120
        grad = self.g0.copy()
        tmp = self.ops.mdot(s, self.hvecs)
        for i in range(self.N):
            scale = 0.0
            for j in range(self.N):
                scale += self.M[i, j] * tmp[j]
            grad += scale * self.hvecs[i]
        return grad


class EigenQuasiNewton:  # :122-291
    """B = B_qn - z0 * hvecs M hvecs^T as ONE compact matrix [Z_qn | hvecs]."""

    def __init__(self, qn, eigh, index=0):
        self.qn, self.eigh, self.index = qn, eigh, index
        self.use_qn_objective = 1
        self.z0 = 1.0
        self.ops = eigh.ops

    def reset(self):
        if self.qn is not None:
            self.qn.reset()

    def update(self, s, y):  # update(x, z, zw, s, y) :176-179
        return 0

    def update_mult(self, x, z, zw):  # :181-187
        self.z0 = float(z[self.index])
        return 0

    def mult(self, x):  # :189-196
        if self.qn is not None and self.use_qn_objective:
            y = self.qn.mult(x)
        else:
            y = np.zeros_like(x)
        return self.eigh.mult_add(-self.z0, x, y)

    def mult_add(self, alpha, x, y):  # :198-204
        if self.qn is not None and self.use_qn_objective:
            self.qn.mult_add(alpha, x, y)
        return self.eigh.mult_add(-alpha * self.z0, x, y)

    @property
    def b0(self):
        return self.get_compact()[0]

    def get_compact(self):  # :212-280
        N = self.eigh.N
        b0, d, Z = 0.0, [], []
        k = 0
        M0 = np.zeros((0, 0))
        if self.qn is not None and self.use_qn_objective:
            b0, d0, M0, Z0 = self.qn.get_compact()
            k = len(Z0)
            d, Z = list(d0[:k]), list(Z0)
        M = np.zeros((k + N, k + N))
        M[:k, :k] = np.asarray(M0)[:k, :k]
        z0inv = 1.0 / self.z0 if self.z0 != 0.0 else 1.0
        M[k:, k:] = z0inv * self.eigh.Minv
        return b0, np.array(d + [1.0] * N), M, Z + list(self.eigh.hvecs)


class EigenSubproblem(_SubproblemBase):  # :295-724
    def __init__(self, prob, approx, update_model=None):
        super().__init__(prob)
        self.approx = approx
        self.update_model = update_model  # callable(x, eigh)

    def get_quasi_newton(self):
        return self.approx

    def init_model_and_bounds(self, tr_size):  # :412-439
        x, lb, ub = self.prob.vars_and_bounds()
        self.xk, self.lb, self.ub = x.copy(), lb.copy(), ub.copy()
        self.set_trust_region_bounds(tr_size)
        _, self.fk, ck = self.prob.eval_obj_con(self.xk)
        self.ck = np.array(ck, dtype=float)
        _, self.gk, self.Ak = self.prob.eval_obj_con_gradient(self.xk)
        if self.update_model is not None:
            eigh = self.approx.eigh
            eigh.c0 = float(self.ck[self.approx.index])
            eigh.g0 = self.Ak[self.approx.index].copy()
            self.update_model(self.xk, eigh)

    def eval_trial_step_and_update(self, update_flag, step, z, zw):  # :460-476
        xtemp = self.xk.copy()
        xtemp += 1.0 * step
        fail, self.ft, ct = self.prob.eval_obj_con(xtemp)
        self.ct = np.array(ct, dtype=float)
        _, self.gt, self.At = self.prob.eval_obj_con_gradient(xtemp)
        return fail, self.ft, self.ct.copy()

    def accept_trial_step(self, step, z, zw):  # :478-529
        xtemp = self.xk.copy()
        xtemp += 1.0 * step
        if self.update_model is not None:
            eigh = self.approx.eigh
            eigh.c0 = float(self.ct[self.approx.index])
            eigh.g0 = self.At[self.approx.index].copy()
            self.update_model(xtemp, eigh)
        qn = self.approx.qn
        if qn is not None:
            t = self._lagrangian_gradient_difference(z, zw)
            qn.update(step.copy(), t)
        self.fk = self.ft
        self.xk = xtemp
        self.gk = self.gt.copy()
        self.ck = self.ct.copy()
        self.Ak = [a.copy() for a in self.At]

    def eval_obj_con(self, step):  # :585-621
        idx = self.approx.index
        eigh = self.approx.eigh
        if step is None:
            cons = self.ck.copy()
            cons[idx] = eigh.eval_approximation(None)
            return 0, self.fk, cons
        fobj = self.fk + self.ops.dot(self.gk, step)
        t = self.approx.mult(step)
        fobj += 0.5 * self.ops.dot(step, t)
        cons = np.zeros(self.c)
        cons[idx] = eigh.eval_approximation(step)
        for i in range(self.c):
            if i != idx:
                cons[i] = self.ck[i] + self.ops.dot(self.Ak[i], step)
        return 0, fobj, cons

    def eval_obj_con_gradient(self, step):  # :626-643
        idx = self.approx.index
        Ac = [a.copy() for a in self.Ak]
        Ac[idx] = self.approx.eigh.eval_approximation_gradient(step)
        g = self.approx.mult(step)
        g = g + 1.0 * self.gk
        return 0, g, Ac


# ---- the driver -----------------------------------------------------------------------------------
class TrustRegion:
    def __init__(self, sub, ip, options=None):
        self.sub, self.ip = sub, ip
        self.opt = dict(TR_DEFAULTS)
        if options:
            for k, v in options.items():
                if k not in self.opt:
                    raise KeyError("unknown trust-region option %s" % k)
                self.opt[k] = v
        self.m = sub.c
        self.nineq = sub.c  # SepProblem: every dense constraint is an inequality
        self.penalty_gamma = np.full(self.m, float(self.opt["penalty_gamma"]))
        self.tr_size = float(self.opt["tr_init_size"])
        self.iter_count = 0
        self.subproblem_iters = 0
        self.adaptive_subproblem_iters = 0
        self.ops = sub.ops
        self.trace = []
        self.hook = None  # hook(self, i) where the reference calls subproblem->writeOutput

    def _infeas(self, c, weights=None):
        tot = 0.0
        for i in range(self.m):
            v = max(0.0, -c[i]) if i < self.nineq else abs(c[i])
            tot += v if weights is None else weights[i] * v
        return tot

    def compute_kkt_error(self, z, zw):  # :2391-2472
        s = self.sub
        relax = self.opt["tr_bound_relax"]
        t = s.gk.copy()
        for i in range(self.m):
            t += -z[i] * s.Ak[i]
        if s.nwcon > 0:
            s.add_sparse_jacobian_transpose(-1.0, zw, t)
        w = t.copy()
        w[(s.xk <= s.lb + relax) & (t > 0.0)] = 0.0
        hi = (s.xk >= s.ub - relax) & (t < 0.0) & ~((s.xk <= s.lb + relax) & (t > 0.0))
        w[hi] = 0.0
        out = self.sub.comm.allreduce([float(np.sum(np.abs(w)))])
        l1 = float(out[0])
        linf = float(self.sub.comm.allreduce([float(np.max(np.abs(w))) if w.size else 0.0], "max")[0])
        zmax = self.ops.maxabs(zw) if s.nwcon > 0 else 0.0
        for i in range(self.m):
            zmax = max(zmax, abs(z[i]))
        zmax = max(1.0, zmax)
        return l1 / max(self.ops.l1norm(s.gk), zmax), linf / max(self.ops.maxabs(s.gk), zmax)

    # ---- filter :896-966 ---------------------------------------------------------------------
    def acceptable_by_pair(self, f_new, h_new, f_old, h_old):
        gamma = self.opt["filter_gamma"]
        if self.opt["filter_sufficient_reduction"]:
            _h_old = (1.0 - gamma) * h_old
            _f_old = f_old - gamma * h_new
        else:
            _h_old, _f_old = h_old, f_old
        return 1 if (h_new < _h_old or f_new < _f_old) else 0

    def acceptable_by_filter(self, f, h):
        for fe, he in self.filter:
            if not self.acceptable_by_pair(f, h, fe, he):
                return 0
        return 1

    def add_to_filter(self, f, h):
        self.filter = [(fe, he) for fe, he in self.filter if not (f <= fe and h <= he)]
        self.filter.append((f, h))

    def minimize_infeas(self, infeas_problem, want_best=True):  # :1105-1228
        o = self.opt
        ip = self.ip
        ipo = ip.opt
        start_option, barrier_option = ipo["starting_point_strategy"], ipo["barrier_strategy"]
        ip.reset_problem_instance(infeas_problem)
        if o["tr_steering_barrier_strategy"] != "default":
            ipo["barrier_strategy"] = o["tr_steering_barrier_strategy"]
        if o["tr_steering_starting_point_strategy"] != "default":
            ipo["starting_point_strategy"] = o["tr_steering_starting_point_strategy"]
        qn = self.sub.get_quasi_newton()
        eig_qn = qn if isinstance(qn, EigenQuasiNewton) else None
        is_seq = ipo["sequential_linear_method"]
        if infeas_problem.objective in ("linear_objective", "constant_objective"):
            if eig_qn is not None:
                eig_qn.use_qn_objective = 0
            if infeas_problem.constraint == "linear_constraint":
                ipo["sequential_linear_method"] = 1
        gamma = 1e6
        if 1e2 * o["tr_penalty_gamma_max"] > gamma:
            gamma = 1e2 * o["tr_penalty_gamma_max"]
        infeas_problem.obj_scale = 1.0 / gamma
        ip.set_penalty_gamma(1.0)
        ip.reset_design_and_bounds()
        ip.optimize()
        step = ip.vars.x.copy()
        if o["tr_accept_step_strategy"] == "penalty_method" and o["tr_adaptive_gamma_update"]:
            self.adaptive_subproblem_iters = ip.niter
        best = None
        if want_best:
            _, _, best = self.sub.eval_obj_con(step)
            best = np.array([max(0.0, -best[j]) if j < self.nineq else abs(best[j]) for j in range(self.m)])
        ip.set_penalty_gamma(self.penalty_gamma)
        ip.reset_problem_instance(self.sub)
        if eig_qn is not None:
            eig_qn.use_qn_objective = 1
        ipo["starting_point_strategy"] = start_option
        ipo["barrier_strategy"] = barrier_option
        ipo["sequential_linear_method"] = is_seq
        return best

    def sl1qp_update(self, step, z, zw):  # :1231-1443
        o = self.opt
        s = self.sub
        _, fk, ck = s.eval_obj_con(None)
        infeas_k = self._infeas(ck, self.penalty_gamma)
        _, ft, ct = s.eval_obj_con(step)
        obj_reduc = fk - ft
        infeas_model = self._infeas(ct, self.penalty_gamma)
        _, ft, ct = s.eval_trial_step_and_update(1, step, z, zw)
        infeas_t = self._infeas(ct, self.penalty_gamma)
        actual_reduc = fk - ft + (infeas_k - infeas_t)
        model_reduc = obj_reduc + (infeas_k - infeas_model)
        fp = o["function_precision"]
        if abs(model_reduc) <= fp and abs(actual_reduc) <= fp:
            rho = 1.0
        else:
            rho = actual_reduc / model_reduc
        infeas = self._infeas(ct)
        accepted = 0
        if rho >= o["tr_eta"] or self.tr_size <= o["tr_min_size"]:
            smax = self.ops.maxabs(step)
            s.accept_trial_step(step, z, zw)
            accepted = 1
        else:
            s.reject_trial_step()
            smax = 0.0
        if rho < 0.25:
            self.tr_size = max(0.25 * self.tr_size, o["tr_min_size"])
        elif rho > 0.75:
            self.tr_size = min(1.5 * self.tr_size, o["tr_max_size"])
        s.set_trust_region_bounds(self.tr_size)
        l1, linfty = self.compute_kkt_error(z, zw)
        toks = []
        if s.qn_update_type == 1:
            toks.append("dampH")
        elif s.qn_update_type == 2:
            toks.append("skipH")
        if o["tr_adaptive_gamma_update"]:
            toks.append("%d/%d" % (self.subproblem_iters, self.adaptive_subproblem_iters))
        else:
            toks.append("%d" % self.subproblem_iters)
        if not accepted:
            toks.append("rej")
        self.trace.append(dict(iter=self.iter_count, fobj=fk, infeas=infeas, l1=l1, linfty=linfty, smax=smax,
                               tr=self.tr_size, rho=rho, model_reduc=model_reduc,
                               zav=float(np.mean(np.abs(z))) if self.m else 0.0,
                               zmax=float(np.max(np.abs(z))) if self.m else 0.0,
                               gav=float(np.mean(self.penalty_gamma)) if self.m else 0.0,
                               gmax=float(np.max(self.penalty_gamma)) if self.m else 0.0, info=toks))
        self.iter_count += 1
        return infeas, l1, linfty

    def filter_optimize(self):  # :1690-2210
        o = self.opt
        ip = self.ip
        s = self.sub
        qn = s.get_quasi_newton()
        ip.set_quasi_newton(qn)
        ip.opt["use_quasi_newton_update"] = 0
        ip.set_penalty_gamma(self.penalty_gamma)
        infeas_problem = InfeasSubproblem(s, "linear_objective", "linear_constraint")
        s.init_model_and_bounds(self.tr_size)
        self.iter_count = 0
        _, f_init, c_init = s.eval_obj_con(None)
        infeas_init = self._infeas(c_init)
        self.filter = []
        self.add_to_filter(-1e20, max(1e4, 1.25 * infeas_init))
        this_resto = last_resto = 0
        tol = o["tr_infeas_tol"]
        for iteration in range(o["tr_max_iterations"]):
            _, fk, ck = s.eval_obj_con(None)
            hk = self._infeas(ck)
            ip.reset_problem_instance(s)
            ip.opt["sequential_linear_method"] = 0
            ip.reset_design_and_bounds()
            ip.optimize()
            step = ip.vars.x.copy()
            z = ip.vars.z.copy()
            zw = ip.vars.zw.copy()
            if o["filter_has_feas_restore_phase"]:
                _, _, cm = s.eval_obj_con(step)
                infeas = 0.0
                for i in range(self.m):  # only the LAST constraint survives this loop (:1826-1832)
                    infeas = max(0.0, abs(-cm[i])) if i < self.nineq else abs(cm[i])
                if infeas > tol:
                    this_resto = 1
                    self.add_to_filter(fk, hk)
                else:
                    this_resto = 0
                    if last_resto:
                        qn.reset()
            if this_resto:
                if not last_resto:
                    qn.reset()
                self.minimize_infeas(infeas_problem, want_best=False)
                # `step`, `z`, `zw` alias the solver's own storage (:1797-1799), so they now hold the
                # restoration LP's solution
                step = ip.vars.x.copy()
                z = ip.vars.z.copy()
                zw = ip.vars.zw.copy()
            _, fobj_model, _ = s.eval_obj_con(step)
            _, fobj_trial, con_trial = s.eval_trial_step_and_update(1, step, z, zw)
            infeas_trial = self._infeas(con_trial)
            smax = self.ops.maxabs(step)
            init_tr = inc_tr = dec_tr = 0
            accepted = 0
            rej = ""
            model_red = fk - fobj_model
            actual_red = fk - fobj_trial
            rho = actual_red / model_red if model_red != 0.0 else float("inf") * (1 if actual_red >= 0 else -1)
            if this_resto:
                s.accept_trial_step(step, None, None)
                accepted = 1
                if smax >= 0.99 * self.tr_size:
                    inc_tr = 1
            else:
                by_filter = self.acceptable_by_filter(fobj_trial, infeas_trial)
                by_pair = self.acceptable_by_pair(fobj_trial, infeas_trial, fk, hk)
                if by_filter and by_pair:
                    if actual_red < o["tr_eta"] * model_red and model_red > 0.0:
                        s.reject_trial_step()
                        smax = 0.0
                        dec_tr = 1
                        rej = "rej:rho"
                    else:
                        s.accept_trial_step(step, None, None)
                        accepted = 1
                        if model_red <= 0.0:
                            self.add_to_filter(fobj_trial, infeas_trial)
                        init_tr = 1
                elif self.tr_size <= o["tr_min_size"]:
                    s.accept_trial_step(step, None, None)
                    accepted = 1
                    if smax >= 0.99 * self.tr_size:
                        inc_tr = 1
                else:
                    s.reject_trial_step()
                    smax = 0.0
                    dec_tr = 1
                    rej = "rej:" + ("" if by_filter else "F") + ("" if by_pair else "xk")
            if self.hook is not None:
                self.hook(self, iteration)
            l1, linfty = self.compute_kkt_error(z, zw)
            toks = []
            if s.qn_update_type == 1:
                toks.append("dampH")
            elif s.qn_update_type == 2:
                toks.append("skipH")
            toks.append("%d" % ip.niter)
            toks.append("f%d" % len(self.filter))
            if this_resto:
                toks.append("R")
            if not accepted:
                toks.append(rej if rej else "rej")
            self.trace.append(dict(iter=self.iter_count, fobj=fobj_trial, infeas=infeas_trial, l1=l1, linfty=linfty,
                                   smax=smax, tr=self.tr_size, rho=rho, model_reduc=model_red,
                                   zav=float(np.mean(np.abs(z))) if self.m else 0.0,
                                   zmax=float(np.max(np.abs(z))) if self.m else 0.0,
                                   gav=float(np.mean(self.penalty_gamma)) if self.m else 0.0,
                                   gmax=float(np.max(self.penalty_gamma)) if self.m else 0.0, info=toks))
            if inc_tr:
                self.tr_size = min(2.0 * self.tr_size, o["tr_max_size"])
            elif dec_tr:
                self.tr_size = max(0.5 * self.tr_size, o["tr_min_size"])
            if init_tr:
                self.tr_size = o["tr_max_size"]
            s.set_trust_region_bounds(self.tr_size)
            self.iter_count += 1
            last_resto = this_resto
            if infeas_trial < tol and (l1 < o["tr_l1_tol"] or linfty < o["tr_linfty_tol"]):
                break
        return 0

    def optimize(self):  # optimize :2365-2384 -> sl1qpOptimize :1453-1687
        o = self.opt
        ip = self.ip
        if o["tr_accept_step_strategy"] == "filter_method":
            return self.filter_optimize()
        ip.set_quasi_newton(self.sub.get_quasi_newton())
        ip.opt["use_quasi_newton_update"] = 0
        ip.set_penalty_gamma(self.penalty_gamma)
        adaptive = o["tr_adaptive_gamma_update"]
        infeas_problem = None
        if adaptive:
            infeas_problem = InfeasSubproblem(self.sub, o["tr_adaptive_objective"], o["tr_adaptive_constraint"])
        self.sub.init_model_and_bounds(self.tr_size)  # initialize() :1086-1099
        self.iter_count = 0
        tol = o["tr_infeas_tol"]
        for i in range(o["tr_max_iterations"]):
            best = None
            if adaptive:
                best = self.minimize_infeas(infeas_problem)
            if self.hook is not None:
                self.hook(self, i)
            ip.reset_design_and_bounds()
            ip.optimize()
            step = ip.vars.x.copy()
            z = ip.vars.z.copy()
            zw = ip.vars.zw.copy()
            self.subproblem_iters = ip.niter
            if adaptive:
                _, _, c0 = self.sub.eval_obj_con(None)
                _, _, cm = self.sub.eval_obj_con(step)
                con_infeas = np.array([max(0.0, -c0[j]) if j < self.nineq else abs(c0[j]) for j in range(self.m)])
                model_infeas = np.array([max(0.0, -cm[j]) if j < self.nineq else abs(cm[j]) for j in range(self.m)])
            infeas, l1, linfty = self.sl1qp_update(step, z, zw)
            if infeas < tol and (l1 < o["tr_l1_tol"] or linfty < o["tr_linfty_tol"]):
                break
            if adaptive:  # :1600-1662
                for j in range(self.m):
                    infeas_reduction = con_infeas[j] - model_infeas[j]
                    best_reduction = con_infeas[j] - best[j]
                    if abs(z[j]) > tol and con_infeas[j] < tol and self.penalty_gamma[j] >= 2.0 * z[j]:
                        self.penalty_gamma[j] = max(0.5 * (self.penalty_gamma[j] + abs(z[j])),
                                                    o["tr_penalty_gamma_min"])
                    elif con_infeas[j] > tol and 0.995 * best_reduction > infeas_reduction:
                        self.penalty_gamma[j] = min(1.5 * self.penalty_gamma[j], o["tr_penalty_gamma_max"])
        return 0
