"""
The reference's Python calling conventions (paropt.ParOpt: Problem(comm, nvars=, ncon=), PVec item
access in the callbacks, Optimizer(problem, options) with algorithm = ip | tr) over the device
library: a dense (non-separable) random quadratic in the style of
examples/random_quadratic/random_quadratic.py, checked against the numpy oracle driven with the
same data.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def make_data(n=40, seed=3):
    rng = np.random.RandomState(seed)
    B = rng.uniform(size=(n, n))
    Q, _, _ = np.linalg.svd(B)
    A = Q @ np.diag(np.linspace(1.0, 50.0, n)) @ Q.T
    return A, rng.uniform(size=n), rng.uniform(size=n), rng.uniform(), -2.0 + rng.uniform(size=n)


def test_reference_style_problem_ip_and_tr(tmp_path):
    from oracle import paropt_oracle as po
    from oracle import tr_oracle as tro
    from paropt_amd import ParOpt

    A, b, Acon, bcon, x0 = make_data()
    n = len(b)

    class Quadratic(ParOpt.Problem):
        def __init__(self):
            self.comm = None
            self.nvars = n
            self.ncon = 1
            super(Quadratic, self).__init__(self.comm, nvars=self.nvars, ncon=self.ncon)

        def getVarsAndBounds(self, x, lb, ub):
            x[:] = x0
            lb[:] = -5.0
            ub[:] = 5.0

        def evalObjCon(self, x):
            con = np.zeros(1, dtype=ParOpt.dtype)
            fobj = 0.5 * np.dot(x, np.dot(A, x)) + np.dot(b, x)
            con[0] = np.dot(x, Acon) + bcon
            return 0, fobj, con

        def evalObjConGradient(self, x, g, Ac):
            g[:] = np.dot(A, x) + b
            Ac[0][:] = Acon[:]
            return 0

    class OracleProblem:  # the same data behind the oracle's problem protocol
        comm = po.SelfComm()
        nlocal, c, nwcon, nwineq = n, 1, 0, 0

        def vars_and_bounds(self):
            return x0.copy(), np.full(n, -5.0), np.full(n, 5.0)

        def eval_obj_con(self, x):
            return 0, 0.5 * x @ (A @ x) + b @ x, np.array([x @ Acon + bcon])

        def eval_obj_con_gradient(self, x):
            return 0, A @ x + b, [Acon.copy()]

    outfile = str(tmp_path / "paropt.out")
    options = {"algorithm": "ip", "abs_res_tol": 1e-8, "starting_point_strategy": "affine_step",
               "barrier_strategy": "monotone", "start_affine_multiplier_min": 0.01, "penalty_gamma": 1000.0,
               "qn_subspace_size": 10, "qn_type": "bfgs", "output_file": outfile}
    opt = ParOpt.Optimizer(Quadratic(), options)
    opt.optimize()
    x, z, zw, zl, zu = opt.getOptimizedPoint()
    oip = po.InteriorPoint(OracleProblem(), {k: v for k, v in options.items() if k not in ("algorithm", "output_file")})
    oip.optimize()
    np.testing.assert_allclose(x[:], oip.vars.x, rtol=0, atol=1e-6)
    np.testing.assert_allclose(z, oip.vars.z, rtol=1e-5, atol=1e-7)
    assert zw is None and len(zl) == n
    names, cols = ParOpt.unpack_output(outfile)
    assert names[0] == "iter" and len(cols[0]) == oip.niter + 1 and cols[1][-1] == oip.neval

    # the same problem through the trust-region driver
    trfile = str(tmp_path / "paropt.tr")
    tr_options = {"algorithm": "tr", "tr_init_size": 0.05, "tr_min_size": 1e-6, "tr_max_size": 10.0, "tr_eta": 0.25,
                  "tr_adaptive_gamma_update": True, "tr_max_iterations": 60, "qn_subspace_size": 10,
                  "output_file": None, "tr_output_file": trfile}
    opt2 = ParOpt.Optimizer(Quadratic(), tr_options)
    opt2.optimize()
    x2 = opt2.getOptimizedPoint()[0]
    ops = po.VecOps(po.SelfComm())
    qn = po.LBFGS(n, 10, ops, "skip_negative_curvature")
    sub = tro.QuadraticSubproblem(OracleProblem(), qn)
    otr = tro.TrustRegion(sub, po.InteriorPoint(sub, {"qn_subspace_size": 10}),
                          {"tr_init_size": 0.05, "tr_min_size": 1e-6, "tr_max_size": 10.0, "tr_eta": 0.25,
                           "tr_max_iterations": 60})
    otr.optimize()
    np.testing.assert_allclose(x2[:], sub.xk, rtol=0, atol=1e-5)
    names, cols = ParOpt.unpack_tr_output(trfile)
    assert len(cols[0]) == otr.iter_count
    # both drivers land on the same optimum of this convex problem
    np.testing.assert_allclose(x2[:], x[:], rtol=0, atol=1e-3)

    # ... and through the method of moving asymptotes (algorithm = "mma")
    from oracle import mma_oracle as mo

    mfile = str(tmp_path / "paropt.mma")
    opt3 = ParOpt.Optimizer(Quadratic(), {"algorithm": "mma", "mma_max_iterations": 12, "mma_output_file": mfile,
                                         "output_file": None})
    opt3.optimize()
    x3, z3 = opt3.getOptimizedPoint()[:2]
    omma = mo.MMA(OracleProblem(), {"mma_max_iterations": 12})
    omma.optimize(po.InteriorPoint(omma, {}))
    np.testing.assert_allclose(x3[:], omma.x, rtol=0, atol=1e-6)
    np.testing.assert_allclose(z3, omma.z, rtol=1e-5, atol=1e-7)
    assert open(mfile).read().count("\n") >= 13


def test_sparse_rosenbrock_example_style():
    """examples/sparse/sparse_rosenbrock.py in spirit: two variables, NO dense constraint, one sparse
    constraint x0 + x1 + 5 >= 0 through the nwblock = 1 callbacks, solved by the interior point and by
    the trust-region driver (the example's own options); checked against the oracle."""
    from oracle import paropt_oracle as po
    from oracle import tr_oracle as tro
    from paropt_amd import ParOpt

    x0 = np.array([-1.3, -0.4])

    class Rosenbrock(ParOpt.Problem):
        def __init__(self):
            self.comm = None
            self.nvars, self.ncon, self.nwcon, self.nwblock = 2, 0, 1, 1
            super(Rosenbrock, self).__init__(self.comm, nvars=self.nvars, ncon=self.ncon, nwcon=self.nwcon,
                                             nwblock=self.nwblock)

        def getVarsAndBounds(self, x, lb, ub):
            x[:] = x0
            lb[:] = -2.0
            ub[:] = 2.0

        def evalObjCon(self, x):
            return 0, 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2, np.zeros(1)

        def evalObjConGradient(self, x, g, A):
            g[0] = 200 * (x[1] - x[0] ** 2) * (-2 * x[0]) - 2 * (1 - x[0])
            g[1] = 200 * (x[1] - x[0] ** 2)
            return 0

        def evalSparseCon(self, x, con):
            con[0] = x[0] + x[1] + 5.0

        def addSparseJacobian(self, alpha, x, px, con):
            con[0] += alpha * (px[0] + px[1])

        def addSparseJacobianTranspose(self, alpha, x, pz, out):
            out[0] += alpha * pz[0]
            out[1] += alpha * pz[0]

        def addSparseInnerProduct(self, alpha, x, c, A):
            A[0] += alpha * (c[0] + c[1])

    class OracleRosen:
        comm = po.SelfComm()
        nlocal, c, nwcon, nwineq = 2, 0, 1, 1

        def vars_and_bounds(self):
            return x0.copy(), np.full(2, -2.0), np.full(2, 2.0)

        def eval_obj_con(self, x):
            return 0, 100 * (x[1] - x[0] ** 2) ** 2 + (1 - x[0]) ** 2, np.zeros(0)

        def eval_obj_con_gradient(self, x):
            return 0, np.array([200 * (x[1] - x[0] ** 2) * (-2 * x[0]) - 2 * (1 - x[0]), 200 * (x[1] - x[0] ** 2)]), []

        def eval_sparse_con(self, x):
            return np.array([x[0] + x[1] + 5.0])

        def add_sparse_jacobian(self, alpha, px, out):
            out[0] += alpha * (px[0] + px[1])
            return out

        def add_sparse_jacobian_transpose(self, alpha, pzw, out):
            out[:] += alpha * pzw[0]
            return out

        def add_sparse_inner_product(self, alpha, cvec, A):
            A[0] += alpha * (cvec[0] + cvec[1])
            return A

    errs = Rosenbrock().checkGradients()  # the example calls it before optimising
    assert errs["objective"] < 1e-4 and errs["transpose"] < 1e-12 and errs["inner_product"] < 1e-12
    ip_opts = {"algorithm": "ip", "qn_subspace_size": 5, "abs_res_tol": 1e-8, "max_major_iters": 200, "output_file": None}
    opt = ParOpt.Optimizer(Rosenbrock(), ip_opts)
    opt.optimize()
    x, z, zw, zl, zu = opt.getOptimizedPoint()
    oip = po.InteriorPoint(OracleRosen(), {"qn_subspace_size": 5, "abs_res_tol": 1e-8, "max_major_iters": 200})
    oip.optimize()
    assert opt.ip.getIterationCounters() == (oip.niter, oip.neval, oip.ngeval)
    np.testing.assert_allclose(x[:], oip.vars.x, rtol=0, atol=1e-7)
    np.testing.assert_allclose(x[:], [1.0, 1.0], atol=1e-5)  # the unconstrained Rosenbrock minimum is feasible
    np.testing.assert_allclose(zw[:], oip.vars.zw, rtol=0, atol=1e-7)

    tr_opts = {"algorithm": "tr", "tr_init_size": 0.5, "tr_min_size": 1e-6, "tr_max_size": 10.0, "tr_eta": 0.1,
               "tr_adaptive_gamma_update": True, "tr_max_iterations": 60, "tr_output_file": None, "output_file": None}
    opt2 = ParOpt.Optimizer(Rosenbrock(), tr_opts)
    opt2.optimize()
    x2 = opt2.getOptimizedPoint()[0]
    ops = po.VecOps(po.SelfComm())
    sub = tro.QuadraticSubproblem(OracleRosen(), po.LBFGS(2, 10, ops, "skip_negative_curvature"))
    otr = tro.TrustRegion(sub, po.InteriorPoint(sub, {}), {"tr_init_size": 0.5, "tr_min_size": 1e-6,
                                                           "tr_max_size": 10.0, "tr_eta": 0.1, "tr_max_iterations": 60})
    otr.optimize()
    assert opt2.tr.getState()["iter_count"] == otr.iter_count
    np.testing.assert_allclose(x2[:], sub.xk, rtol=0, atol=1e-6)


def test_sparse_rosenbrock_csr_form():
    """examples/rosenbrock/sparse_rosenbrock.cpp through the reference's PYTHON interface for the CSR form
    (ParOpt.Problem(comm, ..., rowp=, cols=) with evalSparseObjCon / evalSparseObjConGradient,
    paropt/ParOpt.pyx:579-625,849-881): two dense constraints, nvars - 1 overlapping sparse constraints
    cw_i = 1 - x_i^2 - x_{i+1}^2 >= 0.  Checked against the numpy oracle (pinned to the reference on
    tests/golden/ipcsr_rosenbrock_n100_chain2.npz, the same problem)."""
    from conftest import ip_options_from_case, load_golden
    from oracle import paropt_oracle as po
    from paropt_amd import ParOpt

    n = 100
    rowp = [2 * i for i in range(n)]
    cols = [j for i in range(n - 1) for j in (i, i + 1)]

    class SparseRosenbrock(ParOpt.Problem):
        def __init__(self):
            super(SparseRosenbrock, self).__init__(None, nvars=n, ncon=2, nwcon=n - 1, ninequality=2,
                                                   nwinequality=n - 1, rowp=rowp, cols=cols)

        def getVarsAndBounds(self, x, lb, ub):
            x[:] = -1.0
            lb[:] = -2.0
            ub[:] = 1.0

        def evalSparseObjCon(self, x, sparse):
            xa = np.array(x[:])
            f = np.sum((1.0 - xa[:-1]) ** 2 + 100.0 * (xa[1:] - xa[:-1] ** 2) ** 2)
            con = np.array([0.25 - np.sum(xa * xa), 10.0 + np.sum(xa[::2])])
            sparse[:] = 1.0 - xa[:-1] ** 2 - xa[1:] ** 2
            return 0, f, con

        def evalSparseObjConGradient(self, x, g, A, data):
            xa = np.array(x[:])
            ga = np.zeros(n)
            ga[:-1] += -2.0 * (1.0 - xa[:-1]) + 200.0 * (xa[1:] - xa[:-1] ** 2) * (-2.0 * xa[:-1])
            ga[1:] += 200.0 * (xa[1:] - xa[:-1] ** 2)
            g[:] = ga
            A[0][:] = -2.0 * xa
            A[1][:] = 0.0
            A[1][::2] = 1.0
            data[0::2] = -2.0 * xa[:-1]
            data[1::2] = -2.0 * xa[1:]
            return 0

    errs = SparseRosenbrock().checkGradients(x=np.random.RandomState(0).uniform(-1.5, 0.5, size=n))
    assert errs["objective"] < 1e-3 and errs["con0"] < 1e-4 and errs["con1"] < 1e-6 and errs["transpose"] < 1e-12
    g, case = load_golden("ipcsr_rosenbrock_n100_chain2")
    opts = ip_options_from_case(case)
    opts.pop("write_output_frequency", None)
    opt = ParOpt.Optimizer(SparseRosenbrock(), dict(opts, algorithm="ip", output_file=None))
    opt.optimize()
    x, z, zw, zl, zu = opt.getOptimizedPoint()
    np.testing.assert_array_equal(np.array(opt.ip.getIterationCounters()), g["final/counters"])
    assert abs(opt.ip.getObjective()[0] - g["final/fobj"][0]) <= 1e-6 * abs(g["final/fobj"][0])
    np.testing.assert_allclose(z, g["final/z"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(x[:], g["final/x"], rtol=0, atol=1e-6)
    assert len(zw) == n - 1


def test_electron_style_sparse_equalities_csr():
    """The shape of the reference's examples/COPS/electron/electron.py: NO dense constraint, one sparse EQUALITY
    per electron (x_i^2 + y_i^2 + z_i^2 = 1, num_sparse_inequalities = 0) declared through rowp / cols with the
    reference's keyword names, a dense (non-separable) objective, least-squares multiplier start and damped BFGS.
    Checked against the numpy oracle (dense S) on the same data."""
    from oracle import paropt_oracle as po
    from paropt_amd import ParOpt

    ne = 8
    n = 3 * ne
    rng = np.random.RandomState(0)
    alpha, beta = rng.uniform(0.0, 2 * np.pi, ne), rng.uniform(-np.pi, np.pi, ne)
    x0 = np.concatenate([np.cos(beta) * np.cos(alpha), np.cos(beta) * np.sin(alpha), np.sin(beta)])
    rowp = [3 * i for i in range(ne + 1)]
    cols = [j for i in range(ne) for j in (i, ne + i, 2 * ne + i)]

    def fobj_grad(x):
        P = x.reshape(3, ne)
        d = P[:, :, None] - P[:, None, :]
        dsq = np.sum(d * d, axis=0) + np.eye(ne)
        iu = np.triu_indices(ne, 1)
        f = float(np.sum(dsq[iu] ** -0.5))
        fact = dsq ** -1.5
        np.fill_diagonal(fact, 0.0)
        g = -np.sum(d * fact[None, :, :], axis=2)
        return f, g.reshape(-1)

    class Electron(ParOpt.Problem):
        def __init__(self):
            super(Electron, self).__init__(None, nvars=n, num_sparse_constraints=ne, num_sparse_inequalities=0,
                                           rowp=rowp, cols=cols)

        def getVarsAndBounds(self, x, lb, ub):
            x[:] = x0
            lb[:] = -10.0
            ub[:] = 10.0

        def evalSparseObjCon(self, x, sparse_cons):
            xa = np.array(x[:])
            sparse_cons[:] = 1.0 - np.sum(xa.reshape(3, ne) ** 2, axis=0)
            return 0, fobj_grad(xa)[0], []

        def evalSparseObjConGradient(self, x, g, A, data):
            xa = np.array(x[:])
            g[:] = fobj_grad(xa)[1]
            data[:] = (-2.0 * xa.reshape(3, ne)).T.reshape(-1)
            return 0

    class OracleElectron:
        comm = po.SelfComm()
        nlocal, c, nwcon, nwineq, csr_form = n, 0, ne, 0, True

        def vars_and_bounds(self):
            return x0.copy(), np.full(n, -10.0), np.full(n, 10.0)

        def eval_obj_con(self, x):
            self._cw = 1.0 - np.sum(x.reshape(3, ne) ** 2, axis=0)
            return 0, fobj_grad(x)[0], np.zeros(0)

        def eval_obj_con_gradient(self, x):
            self._A = np.zeros((ne, n))
            for i in range(ne):
                self._A[i, [i, ne + i, 2 * ne + i]] = -2.0 * x[[i, ne + i, 2 * ne + i]]
            return 0, fobj_grad(x)[1], []

        def eval_sparse_con(self, x):
            return self._cw.copy()

        def sparse_jacobian_dense(self):
            return self._A

        def add_sparse_jacobian(self, alpha, px, out):
            out += alpha * (self._A @ px)
            return out

        def add_sparse_jacobian_transpose(self, alpha, pzw, out):
            out += alpha * (self._A.T @ pzw)
            return out

    opts = {"norm_type": "infinity", "qn_type": "bfgs", "qn_subspace_size": 10,
            "starting_point_strategy": "least_squares_multipliers", "qn_update_type": "damped_update",
            "abs_res_tol": 1e-6, "barrier_strategy": "monotone", "armijo_constant": 1e-5, "penalty_gamma": 100.0,
            "max_major_iters": 200}
    opt = ParOpt.Optimizer(Electron(), dict(opts, algorithm="ip", output_file=None))
    opt.optimize()
    x, z, zw, zl, zu = opt.getOptimizedPoint()
    oip = po.InteriorPoint(OracleElectron(), dict(opts))
    oip.optimize()
    assert opt.ip.getIterationCounters() == (oip.niter, oip.neval, oip.ngeval)
    xa = np.array(x[:])
    np.testing.assert_allclose(np.sum(xa.reshape(3, ne) ** 2, axis=0), 1.0, atol=1e-6)  # on the sphere
    np.testing.assert_allclose(xa, oip.vars.x, rtol=0, atol=1e-6)
    np.testing.assert_allclose(np.array(zw[:]), oip.vars.zw, rtol=0, atol=1e-5 * max(1.0, np.abs(oip.vars.zw).max()))


def test_polygon_style_dense_vs_sparse_forms():
    """The shape of the reference's examples/COPS/polygon/polygon.py (largest small polygon): MORE sparse
    constraints than variables (nv-1 ordering rows of 2 entries, nv(nv-1)/2 distance rows of 4), every row
    overlapping many others, so S = C + Aw D^-1 Aw^T is essentially dense (one big front).  The same problem is
    posed (a) with all constraints dense (c = 54 columns in the Gram panel) and (b) in the CSR form; (b) is
    checked against the numpy oracle over the first 60 iterations, (a) and (b) against each other at the end."""
    from oracle import paropt_oracle as po
    from paropt_amd import ParOpt

    nv = 10
    n = 2 * nv
    pairs = [(i, j) for i in range(nv - 1) for j in range(i + 1, nv)]
    nc1, nc2 = nv - 1, len(pairs)
    w = nc1 + nc2
    rng = np.random.RandomState(0)
    x0 = np.concatenate([rng.uniform(0.1, 0.9, nv), np.linspace(0.1 * np.pi, 0.9 * np.pi, nv)])
    lbv = np.zeros(n)
    ubv = np.concatenate([np.full(nv, 10.0), np.full(nv, np.pi)])
    rowp, cols = [0], []
    for i in range(nv - 1):
        cols += [nv + i, nv + i + 1]
        rowp.append(len(cols))
    for i, j in pairs:
        cols += [i, j, nv + i, nv + j]
        rowp.append(len(cols))
    pi_, pj_ = np.array([p[0] for p in pairs]), np.array([p[1] for p in pairs])

    def fobj_grad(x):
        r, t = x[:nv], x[nv:]
        s, c = np.sin(t[1:] - t[:-1]), np.cos(t[1:] - t[:-1])
        f = -0.5 * np.sum(r[:-1] * r[1:] * s)
        g = np.zeros(n)
        g[:nv - 1] -= 0.5 * r[1:] * s
        g[1:nv] -= 0.5 * r[:-1] * s
        g[nv:n - 1] += 0.5 * r[:-1] * r[1:] * c
        g[nv + 1:] -= 0.5 * r[:-1] * r[1:] * c
        return f, g

    def cons_jac(x):
        r, t = x[:nv], x[nv:]
        cval = np.concatenate([t[1:] - t[:-1],
                               1.0 - r[pi_] ** 2 - r[pj_] ** 2 + 2.0 * r[pi_] * r[pj_] * np.cos(t[pi_] - t[pj_])])
        cc, sc = np.cos(t[pi_] - t[pj_]), np.sin(t[pi_] - t[pj_])
        d2 = np.stack([-2.0 * r[pi_] + 2.0 * r[pj_] * cc, -2.0 * r[pj_] + 2.0 * r[pi_] * cc,
                       -2.0 * r[pi_] * r[pj_] * sc, 2.0 * r[pi_] * r[pj_] * sc], axis=1)
        data = np.concatenate([np.tile([-1.0, 1.0], nc1), d2.reshape(-1)])
        return cval, data

    def dense_rows(data):
        A = np.zeros((w, n))
        for i in range(w):
            A[i, cols[rowp[i]:rowp[i + 1]]] = data[rowp[i]:rowp[i + 1]]
        return A

    class SparsePolygon(ParOpt.Problem):
        def __init__(self):
            super(SparsePolygon, self).__init__(None, nvars=n, num_sparse_constraints=w, rowp=rowp, cols=cols)

        def getVarsAndBounds(self, x, lb, ub):
            x[:], lb[:], ub[:] = x0, lbv, ubv

        def evalSparseObjCon(self, x, sparse_cons):
            xa = np.array(x[:])
            sparse_cons[:] = cons_jac(xa)[0]
            return 0, fobj_grad(xa)[0], []

        def evalSparseObjConGradient(self, x, g, A, data):
            xa = np.array(x[:])
            g[:] = fobj_grad(xa)[1]
            data[:] = cons_jac(xa)[1]
            return 0

    class DensePolygon(ParOpt.Problem):
        def __init__(self):
            super(DensePolygon, self).__init__(None, nvars=n, num_dense_constraints=w)

        def getVarsAndBounds(self, x, lb, ub):
            x[:], lb[:], ub[:] = x0, lbv, ubv

        def evalObjCon(self, x):
            xa = np.array(x[:])
            return 0, fobj_grad(xa)[0], cons_jac(xa)[0]

        def evalObjConGradient(self, x, g, A):
            xa = np.array(x[:])
            g[:] = fobj_grad(xa)[1]
            J = dense_rows(cons_jac(xa)[1])
            for i in range(w):
                A[i][:] = J[i]
            return 0

    class OraclePolygon:
        comm = po.SelfComm()
        nlocal, c, nwcon, nwineq, csr_form = n, 0, w, w, True

        def vars_and_bounds(self):
            return x0.copy(), lbv.copy(), ubv.copy()

        def eval_obj_con(self, x):
            self._cw = cons_jac(x)[0]
            return 0, fobj_grad(x)[0], np.zeros(0)

        def eval_obj_con_gradient(self, x):
            self._A = dense_rows(cons_jac(x)[1])
            return 0, fobj_grad(x)[1], []

        def eval_sparse_con(self, x):
            return self._cw.copy()

        def sparse_jacobian_dense(self):
            return self._A

        def add_sparse_jacobian(self, alpha, px, out):
            out += alpha * (self._A @ px)
            return out

        def add_sparse_jacobian_transpose(self, alpha, pzw, out):
            out += alpha * (self._A.T @ pzw)
            return out

    opts = {"norm_type": "infinity", "qn_type": "bfgs", "qn_subspace_size": 10,
            "starting_point_strategy": "least_squares_multipliers", "qn_update_type": "damped_update",
            "abs_res_tol": 1e-6, "barrier_strategy": "monotone", "armijo_constant": 1e-5, "penalty_gamma": 100.0,
            "max_major_iters": 400}
    # (b) against the oracle over the first 60 iterations (the run needs several hundred, as the reference
    # example's own max_major_iters = 500 says; long trajectories part by round-off)
    short = dict(opts, max_major_iters=60)
    sp = ParOpt.Optimizer(SparsePolygon(), dict(short, algorithm="ip", output_file=None))
    sp.optimize()
    xs = np.array(sp.getOptimizedPoint()[0][:])
    oip = po.InteriorPoint(OraclePolygon(), dict(short))
    oip.optimize()
    assert sp.ip.getIterationCounters() == (oip.niter, oip.neval, oip.ngeval)
    np.testing.assert_allclose(xs, oip.vars.x, rtol=0, atol=1e-6)
    # (a) and (b) to the end
    opts["max_major_iters"] = 500
    sp = ParOpt.Optimizer(SparsePolygon(), dict(opts, algorithm="ip", output_file=None))
    sp.optimize()
    xs = np.array(sp.getOptimizedPoint()[0][:])
    dn = ParOpt.Optimizer(DensePolygon(), dict(opts, algorithm="ip", output_file=None))
    dn.optimize()
    xd = np.array(dn.getOptimizedPoint()[0][:])
    # (the example does not pin a vertex at the origin, so the "area" is that of the fan from the origin and the
    # optimum sits at large radii; what matters here is that both forms stay feasible and agree)
    for xx in (xs, xd):
        assert cons_jac(xx)[0].min() >= -1e-5  # ordered angles, all pairwise distances <= 1
        assert fobj_grad(xx)[0] < fobj_grad(x0)[0]
    print("polygon objective: sparse form %.8f, dense form %.8f" % (fobj_grad(xs)[0], fobj_grad(xd)[0]))
    assert abs(fobj_grad(xs)[0] - fobj_grad(xd)[0]) <= 1e-2 * abs(fobj_grad(xd)[0])


@pytest.mark.parametrize("name", ["ipw_rosenbrock_n240_w60_nwblock3", "ipw_quadratic_n240_c2_w60_nwblock3"])
def test_nwblock_three_matches_reference_golden(name):
    """Sparse constraints in the callback block form with nwblock = 3 (ParOptQuasiDefBlockMat, packed 3 x 3 blocks
    from addSparseInnerProduct) through the reference's Python interface, against the trajectories the reference
    produced on the same problems (tests/golden/ipw_*_nwblock3.npz)."""
    from conftest import ip_options_from_case, load_golden
    from oracle import paropt_oracle as po  # problem data only
    from paropt_amd import ParOpt

    g, case = load_golden(name)
    a = case["args"]
    n, w, B, nw = a["n"], a["nwcon"], a["nwblock"], a["nw"]
    data = po.SepProblem(a["problem"], n, a.get("c", 2), nwcon=w, nw=nw, nwstart=a["nwstart"], nwskip=a["nwskip"],
                         nwblock=B)
    nc = data.c
    Aw = data.sparse_jacobian_dense()  # constant: the constraints are linear

    class Blocked(ParOpt.Problem):
        def __init__(self):
            super(Blocked, self).__init__(None, nvars=n, ncon=nc, nwcon=w, nwblock=B)

        def getVarsAndBounds(self, x, lb, ub):
            x[:], lb[:], ub[:] = data.vars_and_bounds()

        def evalObjCon(self, x):
            return data.eval_obj_con(np.array(x[:]))

        def evalObjConGradient(self, x, g_, A):
            fail, gg, AA = data.eval_obj_con_gradient(np.array(x[:]))
            g_[:] = gg
            for j in range(nc):
                A[j][:] = AA[j]
            return fail

        def evalSparseCon(self, x, con):
            con[:] = 1.0 + Aw @ np.array(x[:])

        def addSparseJacobian(self, alpha, x, px, con):
            con[:] = np.array(con[:]) + alpha * (Aw @ np.array(px[:]))

        def addSparseJacobianTranspose(self, alpha, x, pz, out):
            out[:] = np.array(out[:]) + alpha * (Aw.T @ np.array(pz[:]))

        def addSparseInnerProduct(self, alpha, x, c, A):
            S = (Aw * np.array(c[:])) @ Aw.T
            incr = B * (B + 1) // 2
            for b in range(w // B):
                for j in range(B):
                    for i in range(j + 1):
                        A[b * incr + i + j * (j + 1) // 2] += alpha * S[b * B + i, b * B + j]

    errs = Blocked().checkGradients(x=np.random.RandomState(1).uniform(-1.5, 0.5, size=n))
    assert errs["transpose"] < 1e-12 and errs["inner_product"] < 1e-12
    opts = ip_options_from_case(case)
    opts.pop("write_output_frequency", None)
    opt = ParOpt.Optimizer(Blocked(), dict(opts, algorithm="ip", output_file=None))
    opt.optimize()
    x, z, zw, zl, zu = opt.getOptimizedPoint()
    np.testing.assert_array_equal(np.array(opt.ip.getIterationCounters()), g["final/counters"])
    assert abs(opt.ip.getObjective()[0] - g["final/fobj"][0]) <= 1e-6 * abs(g["final/fobj"][0])
    np.testing.assert_allclose(x[:], g["final/x"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(z, g["final/z"], rtol=1e-5, atol=1e-6)
    assert "MatInfo: nblock: 3" in opt.ip.getHistory()
