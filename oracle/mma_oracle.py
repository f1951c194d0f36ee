"""
TEST INFRASTRUCTURE ONLY -- numpy restatement of the reference's method of moving asymptotes
(ParOptMMA, src/ParOptMMA.cpp:28-1052) in the reference's operation order: the separable rational
subproblem (a ParOptProblem seen by the interior-point solver, with its diagonal Hessian), the
asymptote / move-limit update, the KKT error and the outer loop, quirks included (the argument
order of computeKKTError in optimize(), :364-366).  Pinned against tests/golden/mma_*.npz.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import numpy as np

from . import paropt_oracle as po

MMA_DEFAULTS = dict(  # addDefaultOptions :234-289
    mma_max_iterations=200,
    mma_l1_tol=1e-6,
    mma_linfty_tol=1e-6,
    mma_infeas_tol=1e-5,
    mma_use_constraint_linearization=0,
    mma_asymptote_contract=0.7,
    mma_asymptote_relax=1.2,
    mma_init_asymptote_offset=0.5,
    mma_min_asymptote_offset=0.01,
    mma_max_asymptote_offset=10.0,
    mma_bound_relax=0.0,
    mma_eps_regularization=1e-5,
    mma_delta_regularization=1e-3,
    mma_move_limit=0.2,
)


class MMA:
    """Both the subproblem (problem protocol of oracle/paropt_oracle.py) and the driver."""

    def __init__(self, prob, options=None):
        self.prob = prob
        self.comm = prob.comm
        self.ops = po.VecOps(self.comm)
        self.opt = dict(MMA_DEFAULTS)
        if options:
            for k, v in options.items():
                if k not in self.opt:
                    raise KeyError("unknown MMA option %s" % k)
                self.opt[k] = v
        self.nlocal, self.c = prob.nlocal, prob.c
        self.nwcon = getattr(prob, "nwcon", 0)
        self.nwineq = getattr(prob, "nwineq", 0)
        n, m = self.nlocal, self.c
        self.use_true_mma = 1
        self.mma_iter = 0
        self.subproblem_iter = 0
        x, lb, ub = prob.vars_and_bounds()  # initialize() :131-232
        self.x, self.lb, self.ub = x.copy(), lb.copy(), ub.copy()
        self.x1, self.x2 = np.zeros(n), np.zeros(n)
        self.fobj = 0.0
        self.cons = np.zeros(m)
        self.g = np.zeros(n)
        self.A = [np.zeros(n) for _ in range(m)]
        self.L, self.U = np.zeros(n), np.zeros(n)
        self.alpha, self.beta = np.zeros(n), np.ones(n)
        self.p0, self.q0 = np.zeros(n), np.zeros(n)
        self.pi = [np.zeros(n) for _ in range(m)]
        self.qi = [np.zeros(n) for _ in range(m)]
        self.b = np.zeros(m)
        self.cw = np.zeros(self.nwcon)
        self.z = np.zeros(m)
        self.zw = np.zeros(self.nwcon)
        self.zl, self.zu = np.zeros(n), np.zeros(n)
        self.trace = []

    # ---- driver ------------------------------------------------------------------------------
    def compute_kkt_error(self):  # :406-484 -> (l1, linfty, infeas)
        relax = self.opt["mma_bound_relax"]
        r = self.g.copy()
        for i in range(self.c):
            r += -self.z[i] * self.A[i]
        if self.nwcon > 0:
            self.prob.add_sparse_jacobian_transpose(-1.0, self.zw, r)
        if relax <= 0.0:
            r += -1.0 * self.zl
            r += self.zu
            w = np.abs(r)
        else:
            w = r.copy()
            w[(self.x <= self.lb + relax) & (w > 0.0)] = 0.0
            w[(self.x >= self.ub - relax) & (w < 0.0)] = 0.0
            w = np.abs(w)
        l1 = float(self.comm.allreduce([float(np.sum(w))])[0])
        linf = float(self.comm.allreduce([float(np.max(w)) if w.size else 0.0], "max")[0])
        infeas = float(np.sum(np.abs(np.minimum(0.0, self.cons))))
        return l1, linf, infeas

    def initialize_subproblem(self, xv):  # :523-757
        o = self.opt
        movlim = o["mma_move_limit"]
        self.x2 = self.x1.copy()
        self.x1 = self.x.copy()
        if xv is not None:
            self.x = xv.copy()
        x = self.x
        _, self.fobj, cons = self.prob.eval_obj_con(x)
        self.cons = np.array(cons, dtype=float)
        _, self.g, self.A = self.prob.eval_obj_con_gradient(x)
        if self.nwcon > 0:
            self.cw = self.prob.eval_sparse_con(x)
        l1, linf, infeas = self.compute_kkt_error()
        self.trace.append(dict(iter=self.mma_iter, sub_iter=self.subproblem_iter, fobj=self.fobj, l1=l1, linfty=linf,
                               l1_lambda=float(np.sum(np.abs(self.z))), infeas=infeas))
        lower = np.maximum(self.lb, x - movlim)
        upper = np.minimum(self.ub, x + movlim)
        if self.mma_iter < 2:
            off = o["mma_init_asymptote_offset"]
            self.L = x - off * (upper - lower)
            self.U = x + off * (upper - lower)
        else:
            indc = (x - self.x1) * (self.x1 - self.x2)
            intrvl = np.minimum(np.maximum(upper - lower, 0.01), 100.0)
            fac = np.where(indc < 0.0, o["mma_asymptote_contract"], o["mma_asymptote_relax"])
            Ln = x - fac * (self.x1 - self.L)
            Un = x + fac * (self.U - self.x1)
            Ln = np.minimum(Ln, x - o["mma_min_asymptote_offset"] * intrvl)
            Un = np.maximum(Un, x + o["mma_min_asymptote_offset"] * intrvl)
            Ln = np.maximum(Ln, x - o["mma_max_asymptote_offset"] * intrvl)
            Un = np.minimum(Un, x + o["mma_max_asymptote_offset"] * intrvl)
            self.L, self.U = Ln, Un
        L, U = self.L, self.U
        eps, delta = o["mma_eps_regularization"], o["mma_delta_regularization"]
        self.alpha = np.maximum(np.maximum(lower, 0.9 * L + 0.1 * x), x - 0.5 * (upper - lower))
        self.beta = np.minimum(np.minimum(upper, 0.9 * U + 0.1 * x), x + 0.5 * (upper - lower))
        gpos, gneg = np.maximum(0.0, self.g), np.maximum(0.0, -self.g)
        self.p0 = (U - x) * (U - x) * ((1.0 + delta) * gpos + delta * gneg + eps / (U - L))
        self.q0 = (x - L) * (x - L) * ((1.0 + delta) * gneg + delta * gpos + eps / (U - L))
        if self.use_true_mma:
            for i in range(self.c):
                gp, gn = np.maximum(0.0, -self.A[i]), np.maximum(0.0, self.A[i])
                self.pi[i] = (U - x) * (U - x) * gp
                self.qi[i] = (x - L) * (x - L) * gn
                bi = float(np.sum(self.pi[i] / (U - x) + self.qi[i] / (x - L)))
                self.b[i] = float(self.comm.allreduce([bi])[0])
            self.b = -(self.cons + self.b)
        self.mma_iter += 1

    def optimize(self, ip):  # :318-379
        o = self.opt
        self.use_true_mma = 0 if o["mma_use_constraint_linearization"] else 1
        ip.opt["use_diag_hessian"] = True
        ip.opt["use_line_search"] = False
        if ip.hdiag is None:
            ip.hdiag = np.zeros(self.nlocal)
        self.initialize_subproblem(None)
        ip.reset_design_and_bounds()
        for _ in range(o["mma_max_iterations"]):
            ip.optimize()
            self.z = ip.vars.z.copy()
            self.zw = ip.vars.zw.copy()
            self.zl, self.zu = ip.vars.zl.copy(), ip.vars.zu.copy()
            self.initialize_subproblem(ip.vars.x)
            ip.reset_design_and_bounds()
            # computeKKTError(&infeas, &l1, &linfty): the reference passes the outputs in this order
            # into (l1, linfty, infeas) (:364-366), so the names below hold permuted quantities
            infeas, l1, linfty = self.compute_kkt_error()
            if infeas < o["mma_infeas_tol"] and (l1 < o["mma_l1_tol"] or linfty < o["mma_linfty_tol"]):
                break
        return 0

    # ---- the subproblem as a problem ---------------------------------------------------------
    def vars_and_bounds(self):  # :795-799
        return self.x.copy(), self.alpha.copy(), self.beta.copy()

    def eval_obj_con(self, xv):  # :804-866
        L, U = self.L, self.U
        fv = float(self.comm.allreduce([float(np.sum(self.p0 / (U - xv) + self.q0 / (xv - L)))])[0])
        c = np.zeros(self.c)
        for i in range(self.c):
            if self.use_true_mma:
                ci = float(np.sum(self.pi[i] / (U - xv) + self.qi[i] / (xv - L)))
            else:
                ci = float(np.sum(self.A[i] * (xv - self.x)))
            c[i] = float(self.comm.allreduce([ci])[0])
        if self.use_true_mma:
            c = -(c + self.b)
        else:
            c = c + self.cons
        return 0, fv, c

    def eval_obj_con_gradient(self, xv):  # :871-924
        self.subproblem_iter += 1
        Uinv, Linv = 1.0 / (self.U - xv), 1.0 / (xv - self.L)
        g = Uinv * Uinv * self.p0 - Linv * Linv * self.q0
        if self.use_true_mma:
            Ac = [Linv * Linv * self.qi[i] - Uinv * Uinv * self.pi[i] for i in range(self.c)]
        else:
            Ac = [a.copy() for a in self.A]
        return 0, g, Ac

    def hessian_diag(self, xv, z, zw=None):  # :967-1010
        Uinv, Linv = 1.0 / (self.U - xv), 1.0 / (xv - self.L)
        h = 2.0 * (Uinv**3 * self.p0 + Linv**3 * self.q0)
        if self.use_true_mma:
            for i in range(self.c):
                h += 2.0 * z[i] * (Uinv**3 * self.pi[i] + Linv**3 * self.qi[i])
        return h

    def hvec_product(self, xv, z, px, zw=None):  # :929-962 (objective part only, as the reference)
        Uinv, Linv = 1.0 / (self.U - xv), 1.0 / (xv - self.L)
        return 2.0 * (Uinv**3 * self.p0 + Linv**3 * self.q0) * px

    def eval_sparse_con(self, xv):  # :1015-1021
        out = self.cw.copy()
        self.prob.add_sparse_jacobian(1.0, xv, out)
        self.prob.add_sparse_jacobian(-1.0, self.x, out)
        return out

    def add_sparse_jacobian(self, alpha, px, out):
        return self.prob.add_sparse_jacobian(alpha, px, out)

    def add_sparse_jacobian_transpose(self, alpha, pzw, out):
        return self.prob.add_sparse_jacobian_transpose(alpha, pzw, out)

    def add_sparse_inner_product(self, alpha, cvec, A):
        return self.prob.add_sparse_inner_product(alpha, cvec, A)

    # ParOptMMA::createQuasiDefMat forwards to the wrapped problem (src/ParOptMMA.cpp:776-777)
    @property
    def csr_form(self):
        return bool(getattr(self.prob, "chain", None)) or bool(getattr(self.prob, "csr_form", False))

    def sparse_jacobian_dense(self):
        return self.prob.sparse_jacobian_dense()
