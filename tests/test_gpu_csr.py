"""GPU: the CSR form of the sparse constraints (ParOptSparseProblem + ParOptQuasiDefSparseMat, reference
src/ParOptProblem.cpp:624-816, src/ParOptSparseMat.cpp:234-450) through the C ABI.

  * the quasi-definite solve (device assembly of S, level-scheduled sparse Cholesky, triangular solves)
    against a dense numpy solve of the same system, on chain, block-diagonal, grid and random patterns with
    unsorted columns and empty rows - 1e-10 relative (fp64 direct solves of well-conditioned systems);
  * CSR products against numpy;
  * trajectories: tests/golden/ipcsr_*.npz (reference-run) are covered by test_gpu_ip.py; here a callback
    problem in the reference's Python form (rowp=/cols=, evalSparseObjCon / evalSparseObjConGradient) against
    the built-in chain workload and against the numpy oracle, and the block-diagonal CSR pattern against the
    nwblock = 1 path on the same constraints;
  * size-independent properties at w = 1e6: residual of the quasi-definite system, symmetry of the solve.
"""
import ctypes as C

import numpy as np
import pytest

from csr_helpers import chain_pattern, dense_jacobian, grid_pattern, random_pattern

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import paropt_amd as pa

    c = pa.Context(0)
    yield c
    c.close()


class PatternProblem:
    """Callback problem with a given CSR pattern and FIXED Jacobian entries (linear sparse constraints
    cw = b - Aw x... only the quasi-definite machinery is exercised here)."""

    def __new__(cls, ctx, n, rowp, cols, data):
        import paropt_amd as pa

        class _P(pa.Problem):
            def __init__(self):
                super().__init__(ctx, n, 1, 1, nwcon=len(rowp) - 1, nwinequality=len(rowp) - 1, rowp=rowp,
                                 cols=cols)

            def getVarsAndBounds(self, x, lb, ub):
                x[:] = 0.5
                lb[:] = 0.0
                ub[:] = 1.0

            def evalSparseObjCon(self, x, sparse):
                sparse[:] = 1.0 - dense_jacobian(n, rowp, cols, data) @ x
                return 0, float(np.sum(x * x)), np.array([1.0 - np.sum(x)])

            def evalSparseObjConGradient(self, x, g, A, d):
                g[:] = 2.0 * x
                A[0][:] = -1.0
                d[:] = data
                return 0

        return _P()


def _vec(ctx, arr):
    import paropt_amd as pa

    v = pa.PVec(ctx, len(arr))
    v.from_numpy(np.asarray(arr, dtype=float))
    return v


PATTERNS = {
    "chain2": lambda: (300, *chain_pattern(300, 2, 1)),
    "chain5_rev": lambda: (257, *chain_pattern(257, 5, 2, reverse=True)),
    "block_diag": lambda: (240, *chain_pattern(240, 4, 4)),
    "grid": lambda: (12 * 11, *grid_pattern(12, 11)),
    "grid_fronts": lambda: (26 * 24, *grid_pattern(26, 24)),
    "grid_fronts_big": lambda: (40 * 40, *grid_pattern(40, 40)),
    "random_local": lambda: (400, *random_pattern(400, 250, 5, 1, local=12)),
    "random_global": lambda: (150, *random_pattern(150, 90, 3, 2)),
    "long_rows": lambda: (600, *random_pattern(600, 40, 150, 3)),
    "empty_rows": lambda: (50, np.array([0, 0, 2, 2, 3, 3], dtype=np.intc), np.array([4, 1, 4], dtype=np.intc)),
}


@pytest.mark.parametrize("name", sorted(PATTERNS))
def test_quasidef_solve_matches_dense(ctx, name):
    import paropt_amd as pa

    n, rowp, cols = PATTERNS[name]()
    w = len(rowp) - 1
    rng = np.random.default_rng(5)
    data = rng.uniform(-1.5, 1.5, size=int(rowp[-1]))
    prob = PatternProblem(ctx, n, rowp, cols, data)
    x = _vec(ctx, np.full(n, 0.5))
    g = pa.PVec(ctx, n)
    Ac = pa.PVec(ctx, n)
    # one gradient evaluation uploads the Jacobian entries
    ip = pa.InteriorPoint(prob, {"max_major_iters": 0})
    ip.optimize()
    A = dense_jacobian(n, rowp, cols, data)
    d = rng.uniform(0.2, 3.0, size=n)
    c = rng.uniform(0.05, 2.0, size=w)
    S = np.diag(c) + (A * d) @ A.T
    dv, cv = _vec(ctx, d), _vec(ctx, c)
    pa.quasidef_factor(prob, x, dv, cv)
    np.testing.assert_array_equal(cv.to_numpy(), c)  # the CSR form leaves C alone
    info = pa.quasidef_factor_info(prob)
    assert info and "nnz(L)" in info
    if name.startswith("grid_fronts"):
        assert pa.CsrSymbolic(n, rowp, cols).nfronts > 0
    for with_bw in (True, False):
        bx = rng.standard_normal(n)
        bw = rng.standard_normal(w) if with_bw else None
        yx, yw = pa.PVec(ctx, n), pa.PVec(ctx, w)
        pa.quasidef_apply(prob, x, dv, cv, _vec(ctx, bx), _vec(ctx, bw) if with_bw else None, yx, yw)
        rhs = (bw if with_bw else 0.0) - A @ (d * bx)
        yw_ref = np.linalg.solve(S, rhs) if w else np.zeros(0)
        yx_ref = d * (bx + A.T @ yw_ref)
        scale = max(1.0, np.abs(yw_ref).max() if w else 1.0)
        np.testing.assert_allclose(yw.to_numpy(), yw_ref, rtol=0, atol=1e-10 * scale)
        np.testing.assert_allclose(yx.to_numpy(), yx_ref, rtol=0, atol=1e-10 * max(1.0, np.abs(yx_ref).max()))
    # repeated factorizations are bit-identical (no atomics anywhere)
    yw1 = yw.to_numpy()
    pa.quasidef_factor(prob, x, dv, cv)
    yx2, yw2 = pa.PVec(ctx, n), pa.PVec(ctx, w)
    pa.quasidef_apply(prob, x, dv, cv, _vec(ctx, bx), None, yx2, yw2)
    np.testing.assert_array_equal(yw2.to_numpy(), yw1)


def test_non_spd_is_reported(ctx):
    import paropt_amd as pa
    from paropt_amd.lib import ParOptAMDError

    n, rowp, cols = PATTERNS["chain2"]()
    data = np.ones(int(rowp[-1]))
    prob = PatternProblem(ctx, n, rowp, cols, data)
    pa.InteriorPoint(prob, {"max_major_iters": 0}).optimize()
    x = _vec(ctx, np.full(n, 0.5))
    with pytest.raises(ParOptAMDError, match="pivot.*min C -1.000e\\+01"):
        pa.quasidef_factor(prob, x, _vec(ctx, np.ones(n)), _vec(ctx, np.full(len(rowp) - 1, -10.0)))


def _chain_callback_problem(ctx, kind, n, c, span, stride, reverse, seed=0):
    """The built-in chain workload restated as a Python callback problem in the reference's CSR form."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    ref = po.SepProblem(kind, n, c, seed=seed, chain=(span, stride))
    rowp, cols = chain_pattern(n, span, stride, reverse)

    class _P(pa.Problem):
        def __init__(self):
            super().__init__(ctx, n, ref.c, ref.c, nwcon=ref.nwcon, nwinequality=ref.nwcon, rowp=rowp, cols=cols)

        def getVarsAndBounds(self, x, lb, ub):
            x[:], lb[:], ub[:] = ref.vars_and_bounds()

        def evalSparseObjCon(self, x, sparse):
            fail, f, con = ref.eval_obj_con(x)
            sparse[:] = ref._cw
            return fail, f, con

        def evalSparseObjConGradient(self, x, g, A, data):
            fail, gg, AA = ref.eval_obj_con_gradient(x)
            g[:] = gg
            for j in range(ref.c):
                A[j][:] = AA[j]
            jac = ref._jac[:, ::-1] if reverse else ref._jac
            data[:] = jac.reshape(-1)
            return fail

    return _P(), ref


OPTS = {"abs_res_tol": 1e-8, "starting_point_strategy": "affine_step", "barrier_strategy": "monotone",
        "start_affine_multiplier_min": 0.01, "penalty_gamma": 1000.0, "qn_subspace_size": 6, "qn_type": "bfgs",
        "max_major_iters": 40}


@pytest.mark.parametrize("kind,n,c,span,stride,reverse", [("convex", 500, 3, 3, 2, True),
                                                           ("quadratic", 333, 2, 2, 1, False)])
def test_callback_csr_problem_matches_builtin_and_oracle(ctx, kind, n, c, span, stride, reverse):
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    prob_cb, ref = _chain_callback_problem(ctx, kind, n, c, span, stride, reverse)
    ip1 = pa.InteriorPoint(prob_cb, OPTS)
    ip1.optimize()
    built = pa.SeparableProblem(ctx, kind, n, c).setChain(span, stride, reverse)
    ip2 = pa.InteriorPoint(built, OPTS)
    ip2.optimize()
    assert ip1.getIterationCounters() == ip2.getIterationCounters()
    assert "MatInfo: n " in ip2.getHistory()  # the factor line of the reference's output file (:4767-4774)
    x1, z1 = ip1.getOptimizedPoint()[:2]
    x2, z2 = ip2.getOptimizedPoint()[:2]
    np.testing.assert_allclose(x1.to_numpy(), x2.to_numpy(), rtol=0, atol=1e-9)
    np.testing.assert_allclose(z1, z2, rtol=1e-8, atol=1e-9)
    zw1, zw2 = ip1.getOptimizedSparse()[0].to_numpy(), ip2.getOptimizedSparse()[0].to_numpy()
    np.testing.assert_allclose(zw1, zw2, rtol=0, atol=1e-8 * max(1.0, np.abs(zw2).max()))
    # the numpy oracle (pinned to the reference on tests/golden/ipcsr_*.npz) from the same start
    oip = po.InteriorPoint(po.SepProblem(kind, n, c, chain=(span, stride)), dict(OPTS))
    oip.optimize()
    assert (oip.niter, oip.neval, oip.ngeval) == tuple(ip2.getIterationCounters())
    assert abs(oip.fobj - ip2.getObjective()[0]) <= 1e-7 * max(1.0, abs(oip.fobj))
    np.testing.assert_allclose(x2.to_numpy(), oip.vars.x, rtol=0, atol=1e-6)


def test_block_diagonal_csr_equals_block_path(ctx):
    """Linear weighting constraints posed (a) through the nwblock = 1 path and (b) as a CSR pattern with
    disjoint rows: same constraints, two quasi-definite solvers, same trajectory."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    n, c, nwc, nw = 400, 3, 80, 5
    ref = po.SepProblem("convex", n, c, nwcon=nwc, nw=nw)
    rowp = np.arange(nwc + 1, dtype=np.intc) * nw
    cols = np.arange(nwc * nw, dtype=np.intc)

    class _P(pa.Problem):
        def __init__(self):
            super().__init__(ctx, n, c, c, nwcon=nwc, nwinequality=nwc, rowp=rowp, cols=cols)

        def getVarsAndBounds(self, x, lb, ub):
            x[:], lb[:], ub[:] = ref.vars_and_bounds()

        def evalSparseObjCon(self, x, sparse):
            sparse[:] = ref.eval_sparse_con(x)
            return ref.eval_obj_con(x)

        def evalSparseObjConGradient(self, x, g, A, data):
            fail, gg, AA = ref.eval_obj_con_gradient(x)
            g[:] = gg
            for j in range(c):
                A[j][:] = AA[j]
            data[:] = -1.0
            return fail

    ip1 = pa.InteriorPoint(_P(), OPTS)
    ip1.optimize()
    ip2 = pa.InteriorPoint(pa.SeparableProblem(ctx, "convex", n, c).setWeighting(nwc, nw), OPTS)
    ip2.optimize()
    assert ip1.getIterationCounters() == ip2.getIterationCounters()
    np.testing.assert_allclose(ip1.getOptimizedPoint()[0].to_numpy(), ip2.getOptimizedPoint()[0].to_numpy(),
                               rtol=0, atol=1e-8)
    np.testing.assert_allclose(ip1.getOptimizedSparse()[0].to_numpy(), ip2.getOptimizedSparse()[0].to_numpy(),
                               rtol=0, atol=1e-7)


def test_full_size_chain_properties(ctx):
    """w = 1e6 overlapping constraints: the quasi-definite solve satisfies its own system (residual through
    the CSR products), and an interior-point run makes progress with bit-exact repeatability."""
    import paropt_amd as pa

    n = 1_000_001
    prob = pa.SeparableProblem(ctx, "convex", n, 4).setChain(2, 1)
    assert prob.nwcon == n - 1
    ip = pa.InteriorPoint(prob, dict(OPTS, max_major_iters=6))
    ip.optimize()
    f1 = ip.getObjective()[0]
    zw1 = ip.getOptimizedSparse()[0].to_numpy()
    # K0 (yx, -yw) = (bx, bw) residual, using the device products for Aw and Aw^T
    rng = np.random.default_rng(1)
    x = ip.getOptimizedPoint()[0]
    d = rng.uniform(0.5, 2.0, size=n)
    c = rng.uniform(0.1, 1.0, size=n - 1)
    bx, bw = rng.standard_normal(n), rng.standard_normal(n - 1)
    dv, cv, bxv, bwv = (_vec(ctx, a) for a in (d, c, bx, bw))
    yx, yw = pa.PVec(ctx, n), pa.PVec(ctx, n - 1)
    pa.quasidef_factor(prob, x, dv, cv)
    pa.quasidef_apply(prob, x, dv, cv, bxv, bwv, yx, yw)
    xs = x.to_numpy()
    yxa, ywa = yx.to_numpy(), yw.to_numpy()
    # Aw rows: (-2 x_i, -2 x_{i+1})
    Ayx = -2.0 * (xs[:-1] * yxa[:-1] + xs[1:] * yxa[1:])
    ATyw = np.zeros(n)
    ATyw[:-1] += -2.0 * xs[:-1] * ywa
    ATyw[1:] += -2.0 * xs[1:] * ywa
    r1 = yxa / d - ATyw - bx
    r2 = Ayx + c * ywa - bw
    assert np.abs(r1).max() <= 1e-9 * max(1.0, np.abs(bx).max())
    assert np.abs(r2).max() <= 1e-9 * max(1.0, np.abs(bw).max())
    ip2 = pa.InteriorPoint(pa.SeparableProblem(ctx, "convex", n, 4).setChain(2, 1), dict(OPTS, max_major_iters=6))
    ip2.optimize()
    assert ip2.getObjective()[0] == f1
    np.testing.assert_array_equal(ip2.getOptimizedSparse()[0].to_numpy(), zw1)


def test_mma_over_csr_sparse_constraints(ctx):
    """ParOptMMA over a problem in the CSR form: the MMA subproblem forwards the sparse products and the
    quasi-definite factor / solves to the wrapped problem.  With disjoint rows (span = stride) S is diagonal
    and the numpy oracle's MMA driver (block form) solves the same problem: same iteration counts, same
    point.  With overlapping rows the run must complete; MMA's diagonal Hessian can turn indefinite there
    (negative multipliers), which the sparse Cholesky survives and counts, as the reference's does."""
    import paropt_amd as pa
    from oracle import mma_oracle as mo
    from oracle import paropt_oracle as po

    n, c = 150, 2
    opts = {"mma_max_iterations": 6, "output_file": "", "mma_output_file": ""}
    mma = pa.MMA(pa.SeparableProblem(ctx, "convex", n, c).setChain(2, 2), opts)
    mma.optimize()
    omma = mo.MMA(po.SepProblem("convex", n, c, chain=(2, 2)), {"mma_max_iterations": 6})
    omma.optimize(po.InteriorPoint(omma, {}))
    st = mma.getState()
    # (the total of the subproblem iterations can differ by a unit: one solve leaves its slow phase on a
    # round-off level test, see tests/mma_helpers.py)
    assert st["mma_iter"] == omma.mma_iter and abs(st["subproblem_iter"] - omma.subproblem_iter) <= 2
    np.testing.assert_allclose(mma.getOptimizedPoint()[0].to_numpy(), omma.x, rtol=0, atol=1e-6)
    mma2 = pa.MMA(pa.SeparableProblem(ctx, "convex", n, c).setChain(2, 1), opts)
    mma2.optimize()
    x = mma2.getOptimizedPoint()[0].to_numpy()
    assert np.all(np.isfinite(x)) and mma2.getState()["mma_iter"] >= 6


def test_empty_and_tiny_csr_patterns(ctx):
    """nwcon = 0 through the CSR entry point, and a rank too small to own a chain row."""
    import paropt_amd as pa

    prob = pa.SeparableProblem(ctx, "quadratic", 1, 1).setChain(2, 1)  # one variable: no row fits
    assert prob.nwcon == 0
    ip = pa.InteriorPoint(prob, dict(OPTS, max_major_iters=5))
    ip.optimize()
    assert ip.getOptimizedSparse() is None
    prob2 = pa.SeparableProblem(ctx, "quadratic", 2, 1).setChain(2, 1)  # exactly one row
    assert prob2.nwcon == 1
    ip2 = pa.InteriorPoint(prob2, dict(OPTS, max_major_iters=30))
    ip2.optimize()
    x = ip2.getOptimizedPoint()[0].to_numpy()
    assert 1.0 - np.sum(x * x) >= -1e-6


def test_solution_file_with_csr_constraints(ctx, tmp_path):
    """writeSolutionFile / readSolutionFile carry the sparse multipliers of a CSR problem (the reference's
    layout: header nvars, nwcon, ncon; zw and sw after the design blocks)."""
    import paropt_amd as pa

    n = 64
    prob = pa.SeparableProblem(ctx, "convex", n, 2).setChain(3, 2)
    ip = pa.InteriorPoint(prob, dict(OPTS, max_major_iters=12))
    ip.optimize()
    f = str(tmp_path / "sol.bin")
    ip.writeSolutionFile(f)
    raw = open(f, "rb").read()
    hdr = np.frombuffer(raw[:12], dtype="<i4")
    assert tuple(hdr) == (n, prob.nwcon, 2)
    zw = ip.getOptimizedSparse()[0].to_numpy()
    ip2 = pa.InteriorPoint(pa.SeparableProblem(ctx, "convex", n, 2).setChain(3, 2), dict(OPTS, max_major_iters=0))
    ip2.readSolutionFile(f)
    np.testing.assert_array_equal(ip2.getOptimizedSparse()[0].to_numpy(), zw)
    np.testing.assert_array_equal(ip2.getOptimizedPoint()[0].to_numpy(), ip.getOptimizedPoint()[0].to_numpy())


def test_interior_point_on_grid_pattern_with_fronts(ctx):
    """A full interior-point solve whose sparse Schur complement has dense separator fronts (one linear
    inequality per grid edge) against the numpy oracle with a dense S on the same data."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    nx, ny = 26, 24
    n = nx * ny
    rowp, cols = grid_pattern(nx, ny)
    w = len(rowp) - 1
    assert pa.CsrSymbolic(n, rowp, cols).nfronts > 0
    rng = np.random.default_rng(11)
    data = rng.uniform(0.5, 1.5, size=int(rowp[-1]))
    A = dense_jacobian(n, rowp, cols, data)
    xt = rng.uniform(0.2, 0.8, size=n)
    b = A @ rng.uniform(0.3, 0.6, size=n) + 0.05  # cw = b - A x >= 0 is feasible
    a0 = rng.uniform(0.5, 1.0, size=n)

    def f_g(x):
        return float(np.sum((x - xt) ** 2)), 2.0 * (x - xt)

    class P(pa.Problem):
        def __init__(self):
            super().__init__(ctx, n, 1, 1, nwcon=w, nwinequality=w, rowp=rowp, cols=cols)

        def getVarsAndBounds(self, x, lb, ub):
            x[:], lb[:], ub[:] = 0.5, 0.0, 1.0

        def evalSparseObjCon(self, x, sparse):
            sparse[:] = b - A @ x
            return 0, f_g(x)[0], np.array([0.45 * np.sum(a0) - a0 @ x])

        def evalSparseObjConGradient(self, x, g, Ac, d):
            g[:] = f_g(x)[1]
            Ac[0][:] = -a0
            d[:] = data
            return 0

    class O:
        comm = po.SelfComm()
        nlocal, c, nwcon, nwineq, csr_form = n, 1, w, w, True

        def vars_and_bounds(self):
            return np.full(n, 0.5), np.zeros(n), np.ones(n)

        def eval_obj_con(self, x):
            self._cw = b - A @ x
            return 0, f_g(x)[0], np.array([0.45 * np.sum(a0) - a0 @ x])

        def eval_obj_con_gradient(self, x):
            return 0, f_g(x)[1], [-a0.copy()]

        def eval_sparse_con(self, x):
            return self._cw.copy()

        def sparse_jacobian_dense(self):
            return -A

        def add_sparse_jacobian(self, alpha, px, out):
            out -= alpha * (A @ px)
            return out

        def add_sparse_jacobian_transpose(self, alpha, pzw, out):
            out -= alpha * (A.T @ pzw)
            return out

    opts = dict(OPTS, max_major_iters=60)
    # the Jacobian of cw = b - A x is -A: hand the device the entries of -A
    data_dev = -data

    class P2(P):
        def evalSparseObjConGradient(self, x, g, Ac, d):
            g[:] = f_g(x)[1]
            Ac[0][:] = -a0
            d[:] = data_dev
            return 0

    ip = pa.InteriorPoint(P2(), opts)
    ip.optimize()
    oip = po.InteriorPoint(O(), dict(opts))
    oip.optimize()
    assert tuple(ip.getIterationCounters()) == (oip.niter, oip.neval, oip.ngeval)
    assert abs(ip.getObjective()[0] - oip.fobj) <= 1e-8 * max(1.0, abs(oip.fobj))
    np.testing.assert_allclose(ip.getOptimizedPoint()[0].to_numpy(), oip.vars.x, rtol=0, atol=1e-7)
    zw = ip.getOptimizedSparse()[0].to_numpy()
    np.testing.assert_allclose(zw, oip.vars.zw, rtol=0, atol=1e-6 * max(1.0, np.abs(oip.vars.zw).max()))


def test_quasidef_solve_random_soak(ctx):
    """Thirty random patterns (row lengths, locality, grids with random extra rows) against dense solves: guards
    the corner cases of the scheduler (levels without ordinary rows, fronts next to thin rows, table capacity at
    its limits, empty rows)."""
    import paropt_amd as pa

    rng = np.random.default_rng(2024)
    worst = 0.0
    for trial in range(30):
        kind = trial % 3
        if kind == 0:
            n = int(rng.integers(20, 400))
            w = int(rng.integers(1, 300))
            rowp, cols = random_pattern(n, w, int(rng.integers(1, 9)), int(rng.integers(1 << 30)),
                                        local=int(rng.integers(4, 40)))
        elif kind == 1:
            n = int(rng.integers(20, 200))
            w = int(rng.integers(1, 120))
            rowp, cols = random_pattern(n, w, int(rng.integers(1, 40)), int(rng.integers(1 << 30)))
        else:
            nx, ny = int(rng.integers(4, 30)), int(rng.integers(4, 30))
            n = nx * ny
            rowp, cols = grid_pattern(nx, ny)
            extra_p, extra_c = random_pattern(n, int(rng.integers(0, 30)), 6, int(rng.integers(1 << 30)))
            rowp = np.concatenate([rowp, rowp[-1] + extra_p[1:]]).astype(np.intc)
            cols = np.concatenate([cols, extra_c]).astype(np.intc)
        w = len(rowp) - 1
        if rowp[-1] == 0:
            continue
        data = rng.uniform(-1.5, 1.5, size=int(rowp[-1]))
        prob = PatternProblem(ctx, n, rowp, cols, data)
        pa.InteriorPoint(prob, {"max_major_iters": 0}).optimize()
        A = dense_jacobian(n, rowp, cols, data)
        d = rng.uniform(0.2, 3.0, size=n)
        c = rng.uniform(0.05, 2.0, size=w)
        S = np.diag(c) + (A * d) @ A.T
        x = _vec(ctx, np.full(n, 0.5))
        dv, cv = _vec(ctx, d), _vec(ctx, c)
        pa.quasidef_factor(prob, x, dv, cv)
        bx, bw = rng.standard_normal(n), rng.standard_normal(w)
        yx, yw = pa.PVec(ctx, n), pa.PVec(ctx, w)
        pa.quasidef_apply(prob, x, dv, cv, _vec(ctx, bx), _vec(ctx, bw), yx, yw)
        yw_ref = np.linalg.solve(S, bw - A @ (d * bx))
        err = np.abs(yw.to_numpy() - yw_ref).max() / max(1.0, np.abs(yw_ref).max())
        worst = max(worst, err)
        assert err <= 1e-9, (trial, kind, n, w, err)
        yx_ref = d * (bx + A.T @ yw_ref)
        np.testing.assert_allclose(yx.to_numpy(), yx_ref, rtol=0, atol=1e-9 * max(1.0, np.abs(yx_ref).max()))
    assert worst <= 1e-9


def test_borrowed_jacobian_pointer_survives_the_grouped_fallback(ctx):
    """getSparseJacobianData hands out the value array as a borrowed pointer (reference
    src/ParOptProblem.cpp:689-703: valid for the problem's lifetime).  A chain with stride >= span is recognised as the
    grouped pattern and, at the first gradient evaluation with non-uniform entries, falls back to the general CSR path
    in the middle of optimize() -- the analysis then ADOPTS the value array instead of reallocating it: the pointer a
    caller cached before is still the array the library reads, and it holds the entries of the last evaluation."""
    import paropt_amd as pa
    from paropt_amd import lib as L

    n = 64
    prob = pa.SeparableProblem(ctx, "quadratic", n, 3)
    prob.setChain(2, 3)  # non-uniform start: immediate fall-back

    def borrowed():
        rowp, cols = L.c_int_p(), L.c_int_p()
        data, nnz = C.c_void_p(), C.c_int64()
        L.check(L.lib.po_problem_get_sparse_jacobian_data(prob._h, C.byref(rowp), C.byref(cols), C.byref(data), C.byref(nnz)))
        return data.value, nnz.value

    before, nnz = borrowed()
    assert before and nnz == 2 * ((n - 2) // 3 + 1)
    ip = pa.InteriorPoint(prob, {"qn_subspace_size": 4, "max_major_iters": 3, "write_output_frequency": 0})
    ip.optimize()
    after, nnz2 = borrowed()
    assert (after, nnz2) == (before, nnz)
    # the array behind the cached pointer holds the Jacobian of the last gradient evaluation: -2 x on the pattern
    x = ip.getOptimizedPoint()[0].to_numpy()
    vals = np.empty(nnz)
    L.check(L.lib.po_ctx_synchronize(ctx.handle))
    path = next(line.split()[-1] for line in open("/proc/self/maps") if "libamdhip64" in line)  # the runtime in use
    hip = C.CDLL(path)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(vals.ctypes.data, before, 8 * nnz, 2) == 0  # device -> host
    rows = (n - 2) // 3 + 1
    expect = np.array([-2.0 * x[3 * i + k] for i in range(rows) for k in range(2)])
    np.testing.assert_allclose(vals, expect, rtol=0, atol=1e-14)
