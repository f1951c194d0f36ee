"""
Seeded random differential test: the device path (through the C ABI) against the numpy oracle on problem /
size / option combinations drawn at random -- counters, quasi-Newton sizes and info tokens exactly, mu / objective /
norms to 1e-6 over the first iterations.  The fixed sweep of test_gpu_ip.py covers the corners chosen by hand; this
one covers combinations nobody chose.  PAROPT_SWEEP_CASES=<N> widens the campaign (default 24 cases, seed fixed).
"""
import os
import random

import numpy as np
import pytest

from test_gpu_ip import info_tokens

pytestmark = pytest.mark.gpu

NCASES = int(os.environ.get("PAROPT_SWEEP_CASES", "24"))
SEED = int(os.environ.get("PAROPT_SWEEP_SEED", "20261003"))


SMALL_NS = [1, 2, 3, 63, 64, 65, 127, 128, 129, 255, 257, 511, 513, 700, 1023, 1025, 1500, 2049, 3000]
# several tiles per workgroup in every persistent kernel (768 workgroups x 128 rows in the first solve pass, 256 x 128 in
# the producer/consumer Gram), odd lengths, one tile more than a whole round: the sizes the fixed goldens hold only at
# n = 100 000 / 100 003
LARGE_NS = [32769, 65537, 98305, 100003, 131071, 196613, 262147, 300001, 393217]


def draw(rng, ns=SMALL_NS):
    problem = rng.choice(["convex", "quadratic", "quadratic", "rosenbrock"])
    n = rng.choice(ns)
    if problem == "rosenbrock":
        n = max(n, 8)
        c = 2
    else:
        c = rng.choice([1, 1, 2, 3, 4, 7, 8, 9, 16, 17, 31, 32, 33, 40, 40, 70, 97])  # (70 / 97: blocked Gram, collapsed sums)
    qn = rng.choice(["bfgs", "bfgs", "sr1"])
    if qn == "sr1" and n < 3:
        # one variable: the L-SR1 compact matrix s.y - (y.y / s.y) s.s is zero up to one rounding and the reference
        # divides by that rounding error (finite garbage there, 0 / 0 here and in the oracle): not a parity case
        qn = "bfgs"
    m = rng.choice([1, 2, 3, 5, 8, 10, 13])
    if n < 3:
        m = 1  # (more pairs than variables: collinear pairs, a singular compact matrix and coin-flip skip decisions)
    opts = {"qn_subspace_size": m, "qn_type": qn, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01,
            "max_major_iters": 8 if qn == "sr1" else 12}
    if qn == "bfgs":
        opts["barrier_strategy"] = rng.choice(["monotone", "monotone", "mehrotra", "mehrotra_predictor_corrector",
                                               "complementarity_fraction"])
    opts["norm_type"] = rng.choice(["infinity", "infinity", "l1", "l2"])
    if rng.random() < 0.25:
        opts["starting_point_strategy"] = rng.choice(["least_squares_multipliers", "affine_step", "no_start_strategy"])
    if rng.random() < 0.2:
        opts["use_line_search"] = False
    if rng.random() < 0.2:
        opts["use_backtracking_alpha"] = True
    if rng.random() < 0.2 and qn == "bfgs":
        opts["qn_update_type"] = "damped_update"
    if rng.random() < 0.15:
        opts["qn_diag_type"] = rng.choice(["yty_over_yts", "yts_over_sts", "inner_yty_over_yts", "inner_yts_over_sts"])
    if rng.random() < 0.15:
        opts["sequential_linear_method"] = True
    if rng.random() < 0.15 and problem != "rosenbrock":
        opts["use_diag_hessian"] = True
    if rng.random() < 0.15:
        opts["qn_sigma"] = rng.choice([0.1, 1.0])
    if rng.random() < 0.12:
        opts["hessian_reset_freq"] = rng.choice([2, 3, 5])
    if rng.random() < 0.12:
        opts["use_quasi_newton_update"] = False
    if rng.random() < 0.12:
        opts["min_fraction_to_boundary"] = rng.choice([0.9, 0.99])
    if rng.random() < 0.12:
        opts["armijo_constant"] = rng.choice([1e-3, 0.1])
    if rng.random() < 0.12:
        opts["max_line_iters"] = rng.choice([2, 4])
    if rng.random() < 0.12:
        opts["init_barrier_param"] = rng.choice([1.0, 10.0])
    if rng.random() < 0.12 and opts.get("barrier_strategy", "monotone") == "monotone":
        opts["monotone_barrier_fraction"] = rng.choice([0.1, 0.5])
    if rng.random() < 0.1:
        opts["rel_bound_barrier"] = 0.5
    if rng.random() < 0.1:
        opts["penalty_gamma"] = rng.choice([10.0, 100.0])
    if rng.random() < 0.1 and problem != "rosenbrock" and not opts.get("use_diag_hessian") and \
            not opts.get("sequential_linear_method"):
        opts.update(use_hvec_product=True, gmres_subspace_size=rng.choice([4, 8]), nk_switch_tol=1e3, max_gmres_rtol=1.0)
    extra = {}
    if rng.random() < 0.3:
        extra["seed"] = rng.choice([1, 2, 7])
    if problem == "quadratic" and rng.random() < 0.3:
        extra["eig_max"] = rng.choice([10.0, 1e3, 1e5])
    if rng.random() < 0.15:
        extra["bounds_mode"] = rng.choice([2, 5, 7])
    if rng.random() < 0.12 and "bounds_mode" not in extra and problem != "convex":
        # (not on the convex objective b^2 / (eps + x): without its lower bound it has a pole next to the start, without
        # its upper bound it is unbounded below along x -> inf; both runs are round-off lotteries in the reference too)
        # setVarBoundOptions: one-sided bound multipliers (not together with variables placed ON a bound that then has
        # no multiplier: the reference's own norms are nan there)
        extra["bound_options"] = rng.choice([(1, 0), (0, 1)])
    wt = None
    if n >= 64 and rng.random() < 0.3:  # (Rosenbrock too: the shape of examples/rosenbrock/rosenbrock.cpp:131-184)
        nw = rng.choice([2, 3, 5, 8])
        skip = rng.choice([0, 0, 1, 3])
        start = rng.choice([0, 0, 1, 5])
        nwcon = max(1, (n - start) // (nw + skip) // rng.choice([1, 2]))
        nwineq = rng.choice([nwcon, nwcon, nwcon // 2, 0])
        wt = (nwcon, nw, start, skip, nwineq)
        opts.setdefault("starting_point_strategy", "affine_step")
        opts["penalty_gamma"] = 1000.0
    if wt is None and n >= 63 and rng.random() < 0.15 and n <= 5000:
        # (n <= 5000: the oracle holds the chain Jacobian as a dense matrix)
        # the CSR form (ParOptSparseProblem): overlapping chain constraints, device sparse Cholesky
        extra["chain"] = (rng.choice([2, 3]), rng.choice([1, 2]))
        opts.setdefault("starting_point_strategy", "affine_step")
        opts["penalty_gamma"] = opts.get("penalty_gamma", 1000.0)
        opts.pop("use_hvec_product", None)
    return problem, n, c, opts, wt, extra


def cases():
    rng = random.Random(SEED)
    return [draw(rng) for _ in range(NCASES)]


def large_cases_for(seed, ncases):
    rng = random.Random(seed)
    return [draw(rng, LARGE_NS) for _ in range(ncases)]


def cases_for(seed, ncases):
    """The draws of a given seed, independent of the environment (fixtures: oracle/make_sweep_reference.py)."""
    rng = random.Random(seed)
    return [draw(rng) for _ in range(ncases)]


NLARGE = int(os.environ.get("PAROPT_SWEEP_LARGE_CASES", "12"))


def large_cases():
    rng = random.Random(SEED + 1)
    return [draw(rng, LARGE_NS) for _ in range(NLARGE)]


def _make_ctx():
    """PAROPT_SWEEP_RCCL=1: every reduction of the campaign goes through the RCCL communicator (one rank: ncclAllReduce /
    ncclAllGather on the solver's stream, device-to-host copy) instead of the single-rank shortcut."""
    import paropt_amd as pa

    if os.environ.get("PAROPT_SWEEP_RCCL", "0") != "1":
        return pa.Context(0)
    import ctypes as C

    from paropt_amd.lib import check, lib

    os.environ["PAROPT_AMD_FORCE_RCCL"] = "1"
    c = pa.Context(0)
    buf = (C.c_char * 128)()
    check(lib.po_rccl_unique_id(buf))
    check(lib.po_ctx_comm_init_rccl(c.handle, 0, 1, buf))
    assert c.comm_info()[0] == 1
    return c


@pytest.fixture(scope="module")
def ctx():
    c = _make_ctx()
    yield c
    c.close()


@pytest.mark.parametrize("idx", range(NCASES))
def test_random_case_against_oracle(ctx, idx):
    _compare_case_with_oracle(ctx, idx, cases()[idx])


_BASE = {"abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01, "max_major_iters": 12}
RECOGNISED_CHAINS = [
    # CSR chains with stride >= span are row-disjoint groups: recognised as the grouped pattern (problems.cpp) and run
    # on the group kernels while the Jacobian entries -2 x are uniform (Rosenbrock starts at x = -1), on the general
    # CSR path from the first non-uniform evaluation on.  Draw 175 of the round-5 campaign (seed 808) found `grouped`
    # with an empty group map here.
    ("rosenbrock", 513, 2, dict(_BASE, qn_type="bfgs", qn_subspace_size=10, barrier_strategy="mehrotra_predictor_corrector",
                                norm_type="infinity", sequential_linear_method=True,
                                starting_point_strategy="affine_step", penalty_gamma=1000.0), None, {"chain": (2, 2)}),
    ("rosenbrock", 63, 2, dict(_BASE, qn_type="bfgs", qn_subspace_size=2, barrier_strategy="monotone", norm_type="infinity",
                               starting_point_strategy="affine_step", penalty_gamma=1000.0), None, {"seed": 2, "chain": (2, 2)}),
    ("rosenbrock", 200, 2, dict(_BASE, qn_type="bfgs", qn_subspace_size=5, barrier_strategy="monotone", norm_type="l2"),
     None, {"chain": (3, 3)}),
    ("rosenbrock", 257, 2, dict(_BASE, qn_type="bfgs", qn_subspace_size=3, barrier_strategy="mehrotra", norm_type="l1"),
     None, {"chain": (2, 5)}),
    ("quadratic", 64, 3, dict(_BASE, qn_type="bfgs", qn_subspace_size=4, barrier_strategy="monotone", norm_type="l2"),
     None, {"chain": (2, 3)}),
    # the Hessian of a recognised chain (2 zw on its variables) is taken while the pattern is still light (no
    # transposed index on the device): use_diag_hessian evaluates it at the uniform start, before any fall-back
    ("rosenbrock", 130, 2, dict(_BASE, qn_type="bfgs", qn_subspace_size=4, barrier_strategy="monotone", norm_type="infinity",
                                use_diag_hessian=True, max_major_iters=5), None, {"chain": (2, 2)}),
    ("rosenbrock", 97, 2, dict(_BASE, qn_type="bfgs", qn_subspace_size=3, barrier_strategy="monotone", norm_type="l2",
                               use_diag_hessian=True, max_major_iters=5), None, {"chain": (3, 4)}),
]


# ---- the device against the COMPILED REFERENCE itself on drawn cases ---------------------------------------------
# tests/golden/sweep_reference_s424242_n400.npz: what the unmodified reference did on 400 draws of this generator
# (oracle/make_sweep_reference.py runs oracle/_ref/ref_driver on cases_for(424242, 400) and stores its per-iteration
# counters, quasi-Newton sizes, barrier parameters, objectives, norms and info tokens).  No numpy oracle in between: where
# the oracle-based sweep above reports a draw, this one says whether the device or the oracle left the reference.
FIXTURE_SEED, FIXTURE_N = 424242, 400
# draws on which the DEVICE leaves the reference within the compared window, each looked at (profiles/r05_sweep_reference_fixture.txt):
# a decision of the reference taken on round-off (an Armijo or skip / damp test of an iterate that does not move, a Gram
# matrix of rank << its size); the oracle-based sweep meets the same classes
FIXTURE_KNIFE_EDGE = {
    105: "CSR chain with use_diag_hessian: the diagonal block goes INDEFINITE (min D^-1 = -2e4); the device's sparse "
         "Cholesky reports the breakdown and the step is not finite, the fixture's reference ran the driver-side dense "
         "stand-in for ParOptSparseCholesky (not buildable here: METIS), which factors the indefinite block without a "
         "test -- the part of the CSR row that SURVEY 8f / DESIGN 8 leave unpinned",
}
_fixture_cache = {}


# ... and 40 draws at n = 32 769 ... 393 217 (several tiles per workgroup in every persistent kernel):
# tests/golden/sweep_reference_large_s434343_n40.npz (oracle/make_sweep_reference.py --large)
LARGE_FIXTURE_SEED, LARGE_FIXTURE_N = 434343, 40
LARGE_FIXTURE_KNIFE_EDGE = {}


def _fixture(large=False):
    key = "large" if large else "small"
    if key not in _fixture_cache:
        import json

        seed, nc = (LARGE_FIXTURE_SEED, LARGE_FIXTURE_N) if large else (FIXTURE_SEED, FIXTURE_N)
        g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden",
                                 "sweep_reference_%ss%d_n%d.npz" % ("large_" if large else "", seed, nc)))
        drawn = large_cases_for(seed, nc) if large else cases_for(seed, nc)
        # the generator has not drifted from the one the fixture was made with
        assert json.loads(str(g["cases_repr"])) == [repr(cs) for cs in drawn]
        _fixture_cache[key] = (g, drawn)
    return _fixture_cache[key]


def _run_device(ctx, case):
    import paropt_amd as pa

    problem, n, c, opts, wt, extra = case
    prob = pa.SeparableProblem(ctx, problem, n, c, extra.get("seed", 0), 1.0, extra.get("eig_max", 100.0))
    if wt:
        prob.setWeighting(*wt)
    if extra.get("bounds_mode", 0):
        prob.setBoundsMode(extra["bounds_mode"])
    if extra.get("chain"):
        prob.setChain(*extra["chain"])
    if extra.get("bound_options"):
        prob.setVarBoundOptions(*extra["bound_options"])
    ip = pa.InteriorPoint(prob, dict(opts, write_output_frequency=0))
    gsn = []
    ip.setIterationCallback(lambda k: gsn.append(ip.snapshot()))
    ip.optimize()
    return gsn, info_tokens(ip.getHistory())


@pytest.mark.parametrize("idx", range(FIXTURE_N))
def test_random_case_against_reference_fixture(ctx, idx):
    _compare_with_fixture(ctx, idx, False)


@pytest.mark.parametrize("idx", range(LARGE_FIXTURE_N))
def test_random_large_case_against_reference_fixture(ctx, idx):
    _compare_with_fixture(ctx, idx, True)


def _compare_with_fixture(ctx, idx, large):
    import json

    g, drawn = _fixture(large)
    skips = LARGE_FIXTURE_KNIFE_EDGE if large else FIXTURE_KNIFE_EDGE
    if idx in skips:
        pytest.skip(skips[idx])
    case = drawn[idx]
    problem, n, c, opts, wt, extra = case
    what = (idx,) + tuple(case)
    pre = "d%04d/" % idx
    rc, rq, rmu, rf, rn = (g[pre + k] for k in ("counters", "qn_size", "mu", "fobj", "norms"))
    rtok = {int(k): v for k, v in json.loads(str(g[pre + "tokens"])).items()}
    gsn, gtok = _run_device(ctx, case)
    bopt = extra.get("bound_options")
    # the windows of the oracle-based sweep (see _compare_case_with_oracle)
    ncmp = min(len(rc), len(gsn), 6 if opts["qn_type"] == "sr1" else 8)
    if opts.get("barrier_strategy") == "mehrotra_predictor_corrector":
        ncmp = min(ncmp, 6)
    assert ncmp >= min(len(rc), 4), what
    for k in range(ncmp):
        if k > 2 and float(np.nanmax(rn[k])) < 1e-7:
            ncmp = k
            break
    for k in range(ncmp):
        np.testing.assert_array_equal(gsn[k]["counters"], rc[k], err_msg="counters @%d %r" % (k, what))
        assert gsn[k]["qn_size"] == rq[k], ("qn_size", k, gsn[k]["qn_size"], int(rq[k]), what)
        assert abs(gsn[k]["mu"] - rmu[k]) <= 1e-6 * abs(rmu[k]), ("mu", k, gsn[k]["mu"], float(rmu[k]), what)
        assert abs(gsn[k]["fobj"] - rf[k]) <= 1e-6 * max(1.0, abs(rf[k])), ("fobj", k, gsn[k]["fobj"], float(rf[k]), what)
        gn, on = np.array(gsn[k]["norms"], dtype=float), np.array(rn[k], dtype=float)
        if bopt:
            keep = np.array([True, bool(bopt[0]), bool(bopt[1])])
            gn, on = gn[keep], on[keep]
        np.testing.assert_allclose(gn, on, rtol=1e-6, atol=1e-11, err_msg="norms @%d %r" % (k, what))
    assert [rtok.get(k, []) for k in range(1, ncmp)] == [gtok.get(k, []) for k in range(1, ncmp)], ("tokens", what)


@pytest.mark.parametrize("idx", range(NLARGE))
def test_random_large_case_against_oracle(ctx, idx):
    """The same draws at sizes where every persistent kernel runs several tiles per workgroup (LARGE_NS).  In the suite
    these sizes are held to the compiled reference's fixture (test_random_large_case_against_reference_fixture: data, the
    same on every box); the oracle-based form -- whose numpy sums over 1e5 terms depend on the host's library -- runs on
    request (PAROPT_SWEEP_LARGE_ORACLE=1) and in the campaign (python tests/test_gpu_random_sweep.py)."""
    if os.environ.get("PAROPT_SWEEP_LARGE_ORACLE", "0") != "1" and "PAROPT_SWEEP_LARGE_CASES" not in os.environ:
        pytest.skip("oracle-based large-n sweep: on request (PAROPT_SWEEP_LARGE_ORACLE=1); the suite uses the reference fixture")
    _compare_case_with_oracle(ctx, 20000 + idx, large_cases()[idx])


@pytest.mark.parametrize("k", range(len(RECOGNISED_CHAINS)))
def test_recognised_chain_patterns_against_oracle(ctx, k):
    _compare_case_with_oracle(ctx, 10000 + k, RECOGNISED_CHAINS[k])


@pytest.mark.parametrize("chain", [(2, 2), (3, 5)])
def test_hessian_vector_product_of_a_recognised_chain(ctx, chain):
    """evalHvecProduct / evalHessianDiag of a chain problem whose CSR pattern was recognised as grouped (light
    pattern: d_colp / d_rowsT are never uploaded).  The Newton-Krylov step asks for Hessian-vector products with the
    sparse multipliers at the uniform start, while the problem is still grouped: the column sums 2 zw come from the
    group scatter (the library dereferenced null device pointers before).  Compared with the same problem forced onto
    the general CSR path (PAROPT_AMD_NO_CSR_GROUPS is read once per process, so the comparison is with the oracle)."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    n = 101
    opts = {"qn_type": "bfgs", "qn_subspace_size": 3, "max_major_iters": 3, "use_hvec_product": True,
            "gmres_subspace_size": 4, "nk_switch_tol": 1e3, "max_gmres_rtol": 1.0, "abs_res_tol": 1e-8,
            "start_affine_multiplier_min": 0.01}
    prob = pa.SeparableProblem(ctx, "rosenbrock", n, 2)
    prob.setChain(*chain)
    ip = pa.InteriorPoint(prob, dict(opts, write_output_frequency=0))
    rep = ip.checkGradients(1e-6)
    assert "Hessian-vector product test" in rep, rep
    gsn = []
    ip.setIterationCallback(lambda k: gsn.append(ip.snapshot()))
    ip.optimize()
    oip = po.InteriorPoint(po.SepProblem("rosenbrock", n, 2, chain=chain), opts)
    osn = []
    oip.hook = lambda s, k: osn.append(s.snapshot())
    oip.optimize()
    assert len(gsn) >= 2 and len(gsn) == len(osn)
    for k in range(len(gsn)):
        np.testing.assert_array_equal(gsn[k]["counters"], osn[k]["counters"])
        assert abs(gsn[k]["fobj"] - osn[k]["fobj"]) <= 1e-8 * max(1.0, abs(osn[k]["fobj"])), (k, gsn[k]["fobj"], osn[k]["fobj"])


def _compare_case_with_oracle(ctx, idx, case):
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    problem, n, c, opts, wt, extra = case
    wargs = dict(nwcon=wt[0], nw=wt[1], nwstart=wt[2], nwskip=wt[3], nwineq=wt[4]) if wt else {}
    wargs.update(extra)
    bopt = wargs.pop("bound_options", None)
    oprob = po.SepProblem(problem, n, c, **wargs)
    if bopt:
        oprob.use_lower, oprob.use_upper = bool(bopt[0]), bool(bopt[1])
    oip = po.InteriorPoint(oprob, opts)
    osn = []
    oip.hook = lambda s, k: osn.append(s.snapshot())
    try:
        oip.optimize()
    except np.linalg.LinAlgError:
        # the oracle factors the sparse-constraint block with a dense Cholesky and gives up where the iterate has
        # made it numerically indefinite (drawn CSR chains with non-convergent option sets): nothing to compare with
        pytest.skip("the oracle's dense Cholesky of the sparse-constraint block failed")
    prob = pa.SeparableProblem(ctx, problem, n, c, extra.get("seed", 0), 1.0, extra.get("eig_max", 100.0))
    if wt:
        prob.setWeighting(*wt)
    if extra.get("bounds_mode", 0):
        prob.setBoundsMode(extra["bounds_mode"])
    if extra.get("chain"):
        prob.setChain(*extra["chain"])
    if bopt:
        prob.setVarBoundOptions(*bopt)
    ip = pa.InteriorPoint(prob, dict(opts, write_output_frequency=0))
    gsn = []
    ip.setIterationCallback(lambda k: gsn.append(ip.snapshot()))
    ip.optimize()
    what = (idx, problem, n, c, opts, wt, extra)
    # eight iterations: beyond that, combinations that do not converge (the sequential linear method or the
    # predictor-corrector on the convex objective) take Armijo decisions on a knife's edge -- in three of 400 drawn
    # cases the reference, the oracle and the device path each count a different number of line-search evaluations at
    # iteration 9 or 11
    # (L-SR1 inside the line-search method does not converge at all, SURVEY 8d: three of 1 600 drawn cases differ in the
    # evaluation count of iteration 6 or 7 -- six iterations there)
    ncmp = min(len(osn), len(gsn), 6 if opts["qn_type"] == "sr1" else 8)
    if opts.get("barrier_strategy") == "mehrotra_predictor_corrector":
        ncmp = min(ncmp, 6)  # (the predictor-corrector runs are the ones with the knife-edge decisions at iteration 7)
    assert ncmp >= min(len(osn), 4), what
    # a difference of 1e-16 decides branches once the iterate is at round-off level (converged tiny problems, the
    # non-convergent L-SR1 iteration): the comparison stops where the oracle's residual is below 1e-7
    for k in range(ncmp):
        if k > 2 and float(np.max(osn[k]["norms"])) < 1e-7:
            ncmp = k
            break
    for k in range(ncmp):
        np.testing.assert_array_equal(gsn[k]["counters"], osn[k]["counters"], err_msg="counters @%d %r" % (k, what))
        assert gsn[k]["qn_size"] == osn[k]["qn_size"], (k, what)
        assert abs(gsn[k]["mu"] - osn[k]["mu"]) <= 1e-6 * abs(osn[k]["mu"]), (k, what)
        assert abs(gsn[k]["fobj"] - osn[k]["fobj"]) <= 1e-6 * max(1.0, abs(osn[k]["fobj"])), (k, what)
        gn, on = np.array(gsn[k]["norms"], dtype=float), np.array(osn[k]["norms"], dtype=float)
        if bopt:  # the unused bound multiplier is not handed out by getOptimizedPoint (snapshot: nan); the oracle keeps
            keep = np.array([True, bool(bopt[0]), bool(bopt[1])])  # the reference's untouched initial values there
            gn, on = gn[keep], on[keep]
        np.testing.assert_allclose(gn, on, rtol=1e-6, atol=1e-11, err_msg="%d %r" % (k, what))
        if wt:
            np.testing.assert_allclose(gsn[k]["wnorms"], osn[k]["wnorms"], rtol=1e-6, atol=1e-11)
    assert [t["info"].split() for t in oip.trace[1:ncmp]] == [
        info_tokens(ip.getHistory()).get(k, []) for k in range(1, ncmp)], what




# ---- the same drawn cases through the host-callback boundary ---------------------------------------------------------
def _host_twin(pa, ctx, oprob, n, c, wt):
    """A user problem written on host arrays (getArray views), as the reference's Python examples are: every callback
    delegates to the oracle's problem definition."""

    class Twin(pa.Problem):
        def getVarsAndBounds(self, x, lb, ub):
            x0, l0, u0 = oprob.vars_and_bounds()
            x[:], lb[:], ub[:] = x0, l0, u0

        def evalObjCon(self, x):
            return oprob.eval_obj_con(x)

        def evalObjConGradient(self, x, g, A):
            _, gg, aa = oprob.eval_obj_con_gradient(x)
            g[:] = gg
            for j in range(c):
                A[j][:] = aa[j]
            return 0

        def evalHvecProduct(self, x, z, zw, px, hvec):
            hvec[:] = oprob.hvec_product(x, z, px, zw)
            return 0

        def evalHessianDiag(self, x, z, zw, hdiag):
            hdiag[:] = oprob.hessian_diag(x, z, zw)
            return 0

        def evalSparseCon(self, x, out):
            out[:] = oprob.eval_sparse_con(x)
            return 0

        def addSparseJacobian(self, alpha, x, px, out):
            oprob.add_sparse_jacobian(alpha, px, out)
            return 0

        def addSparseJacobianTranspose(self, alpha, x, pzw, out):
            oprob.add_sparse_jacobian_transpose(alpha, pzw, out)
            return 0

        def addSparseInnerProduct(self, alpha, x, cvec, A):
            oprob.add_sparse_inner_product(alpha, cvec, A)
            return 0

    if wt:
        return Twin(ctx, n, c, nwcon=wt[0], nwinequality=wt[4])
    return Twin(ctx, n, c)


@pytest.mark.parametrize("idx", [i for i in range(NCASES) if i % 3 == 0])
def test_random_case_through_host_callbacks(ctx, idx):
    """Every third drawn case again with the problem implemented in Python on host arrays (the mirror / upload
    machinery of po_vec_get_array under all those option combinations), against the oracle like the device problem."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    problem, n, c, opts, wt, extra = cases()[idx]
    if extra.get("chain") or extra.get("bounds_mode") or extra.get("bound_options"):
        pytest.skip("CSR form / broken bounds / bound options are set-up calls of the built-in problem")
    wargs = dict(nwcon=wt[0], nw=wt[1], nwstart=wt[2], nwskip=wt[3], nwineq=wt[4]) if wt else {}
    wargs.update(extra)
    oip = po.InteriorPoint(po.SepProblem(problem, n, c, **wargs), opts)
    osn = []
    oip.hook = lambda s, k: osn.append(s.snapshot())
    oip.optimize()
    twin = _host_twin(pa, ctx, po.SepProblem(problem, n, c, **wargs), n, c, wt)
    ip = pa.InteriorPoint(twin, dict(opts, write_output_frequency=0))
    gsn = []
    ip.setIterationCallback(lambda k: gsn.append(ip.snapshot()))
    ip.optimize()
    what = ("host", idx, problem, n, c, opts, wt, extra)
    ncmp = min(len(osn), len(gsn), 6 if opts["qn_type"] == "sr1" else 8)
    if opts.get("barrier_strategy") == "mehrotra_predictor_corrector":
        ncmp = min(ncmp, 6)
    assert ncmp >= min(len(osn), 4), what
    for k in range(ncmp):
        if k > 2 and float(np.max(osn[k]["norms"])) < 1e-7:
            ncmp = k
            break
    for k in range(ncmp):
        np.testing.assert_array_equal(gsn[k]["counters"], osn[k]["counters"], err_msg="counters @%d %r" % (k, what))
        assert gsn[k]["qn_size"] == osn[k]["qn_size"], (k, what)
        assert abs(gsn[k]["mu"] - osn[k]["mu"]) <= 1e-6 * abs(osn[k]["mu"]), (k, what)
        assert abs(gsn[k]["fobj"] - osn[k]["fobj"]) <= 1e-6 * max(1.0, abs(osn[k]["fobj"])), (k, what)
        np.testing.assert_allclose(gsn[k]["norms"], osn[k]["norms"], rtol=1e-6, atol=1e-11, err_msg="%d %r" % (k, what))



# ---- the user-side C++ problem (examples/random_convex_amd.cpp) under drawn options ------------------------------------
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
USER_LIB = os.path.join(ROOT, "examples", "librandom_convex_user.so")


def facade_cases():
    out = []
    for i, (problem, n, c, opts, wt, extra) in enumerate(cases()):
        if problem == "convex" and wt is None and not extra and not opts.get("use_hvec_product") \
                and not opts.get("use_diag_hessian") and n >= 2:
            out.append(i)
    return out


@pytest.mark.parametrize("idx", facade_cases())
def test_random_case_user_library_problem_matches_builtin(ctx, idx):
    """The drawn convex cases once more with the problem living OUTSIDE the library (a ParOptProblem subclass on the C++
    facade with its own kernels): counters, info tokens and pivots as the built-in twin, state to round-off."""
    import paropt_amd as pa

    assert os.path.exists(USER_LIB), "examples/librandom_convex_user.so is not built (__graft_entry__.build())"
    problem, n, c, opts, wt, extra = cases()[idx]
    runs = []
    for make in (lambda: pa.SeparableProblem(ctx, "convex", n, c), lambda: pa.UserLibraryProblem(ctx, USER_LIB, n, c)):
        prob = make()
        ip = pa.InteriorPoint(prob, dict(opts, write_output_frequency=0))
        sn = []
        ip.setIterationCallback(lambda k: sn.append(ip.snapshot()))
        ip.optimize()
        runs.append((sn, info_tokens(ip.getHistory()), ip.getIterationCounters()))
        del ip
        if hasattr(prob, "close"):
            prob.close()
    (sa, ta, ca), (sb, tb, cb) = runs
    what = ("facade", idx, n, c, opts)
    ncmp = min(len(sa), len(sb), 6 if opts["qn_type"] == "sr1" else 8)
    if opts.get("barrier_strategy") == "mehrotra_predictor_corrector":
        ncmp = min(ncmp, 6)
    for k in range(ncmp):
        np.testing.assert_array_equal(sa[k]["counters"], sb[k]["counters"], err_msg=repr((k, what)))
        assert sa[k].get("qn_size", 0) == sb[k].get("qn_size", 0), (k, what)
        assert abs(sa[k]["mu"] - sb[k]["mu"]) <= 1e-8 * abs(sa[k]["mu"]), (k, what)
        assert abs(sa[k]["fobj"] - sb[k]["fobj"]) <= 1e-8 * max(1.0, abs(sa[k]["fobj"])), (k, what)
        assert ta.get(k, []) == tb.get(k, []), (k, what)


# ---- compact quasi-Newton classes -----------------------------------------------------------------------------------
NQN = int(os.environ.get("PAROPT_SWEEP_QN_CASES", "16"))


def qn_cases():
    rng = random.Random(SEED + 2)
    out = []
    for _ in range(NQN):
        kind = rng.choice(["bfgs", "bfgs", "sr1"])
        n = rng.choice([1, 2, 3, 7, 63, 64, 65, 127, 129, 255, 300, 511, 513, 1025, 2050])
        if kind == "sr1" and n < 3:
            kind = "bfgs"  # (one variable under L-SR1: the compact matrix is zero up to one rounding, see draw())
        m = rng.choice([1, 2, 3, 5, 8, 12, 20])
        update = rng.choice(["skip_negative_curvature", "damped_update"])
        diag = rng.choice(["yty_over_yts", "yts_over_sts", "inner_yty_over_yts", "inner_yts_over_sts"])
        steps = rng.choice([3, m + 2, 2 * m + 3])
        # kinds of pairs: 0 well-behaved, 1 negative curvature, 2 tiny step, 3 huge y, 4 y = 0
        seq = [rng.choice([0, 0, 0, 0, 1, 2, 3, 4]) for _ in range(steps)]
        out.append((kind, n, m, update, diag, seq, rng.randrange(1 << 30)))
    return out


@pytest.mark.parametrize("idx", range(NQN))
def test_random_quasi_newton_sequence_against_oracle(ctx, idx):
    """Drawn update sequences (well-behaved, negative-curvature, tiny, huge and zero pairs; widths from 1 to 20;
    vector lengths around the tile sizes) through the device L-BFGS / L-SR1 and through the oracle: return codes and
    compact-matrix sizes exactly, b0 / d0 / M and the products to the tolerances of the golden test."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    kind, n, m, update, diag, seq, seed = qn_cases()[idx]
    what = qn_cases()[idx]
    ops = po.VecOps(po.SelfComm())
    if kind == "bfgs":
        qn, oq = pa.LBFGS(ctx, n, m, update), po.LBFGS(n, m, ops, update)
    else:
        qn, oq = pa.LSR1(ctx, n, m), po.LSR1(n, m, ops)
    qn.setInitDiagonalType(diag)
    oq.diag_type = diag
    rng = np.random.default_rng(seed)
    h = 0.5 + 4.0 * rng.random(n)
    xp_np = rng.standard_normal(n)
    xp, s, y, out = (pa.PVec(ctx, n) for _ in range(4))
    xp.from_numpy(xp_np)
    for k, pk in enumerate(seq):
        sn = rng.standard_normal(n)
        yn = h * sn + 0.1 * rng.standard_normal(n)
        if pk == 1:
            yn = -0.5 * h * sn
        elif pk == 2:
            sn, yn = 1e-9 * sn, 1e-9 * yn
        elif pk == 3:
            yn = 1e8 * yn
        elif pk == 4:
            yn = np.zeros(n)
        s.from_numpy(sn)
        y.from_numpy(yn)
        rc, orc = qn.update(s, y), oq.update(sn.copy(), yn.copy())
        assert rc == orc, (k, what)
        b0, d0, M, Z = qn.getCompactMat()
        ob0, od0, oM, oZ = oq.get_compact()
        assert len(Z) == len(oZ), (k, what)
        assert abs(b0 - ob0) <= 1e-9 * abs(ob0), (k, b0, ob0, what)  # (3e-10 after forty damped updates with 1e8-scaled pairs)
        if len(Z) and np.all(np.isfinite(oM)) and np.all(np.isfinite(od0)):
            np.testing.assert_allclose(d0, od0, rtol=1e-9, err_msg=repr((k, what)))
            np.testing.assert_allclose(M, oM, rtol=1e-8, atol=1e-9 * max(1e-300, np.abs(oM).max()), err_msg=repr((k, what)))
            want = oq.mult(xp_np)
            if np.all(np.isfinite(want)) and np.linalg.cond(oM) < 1e10:
                qn.mult(xp, out)
                # (floor: with nothing but zero pairs in memory the product is round-off of size 1e-16 on both sides)
                np.testing.assert_allclose(out.to_numpy(), want, rtol=0, atol=1e-6 * max(1e-9, np.abs(want).max()),
                                           err_msg=repr((k, what)))


if __name__ == "__main__":  # python tests/test_gpu_random_sweep.py: the campaign with one line per failing case
    os.environ.setdefault("PAROPT_SWEEP_LARGE_ORACLE", "1")  # (the campaign runs the oracle-based large-n draws)
    c = _make_ctx()
    nbad = 0
    for i in range(NCASES):
        try:
            test_random_case_against_oracle(c, i)
        except AssertionError as e:
            nbad += 1
            msg = str(e).strip().splitlines()
            print("CASE %d %r\n     -> %s" % (i, cases()[i], " | ".join(m.strip() for m in msg[:6])[:700]), flush=True)
        except BaseException as e:  # pytest.skip
            if type(e).__name__ == "Skipped":
                continue
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                raise
            nbad += 1  # an error of the library is a finding of the campaign like any other: report it and go on
            print("CASE %d %r\n     -> ERROR %s: %s" % (i, cases()[i], type(e).__name__, str(e)[:500]), flush=True)
    print("%d of %d cases differ" % (nbad, NCASES))
    if os.environ.get("PAROPT_SWEEP_FIXTURE", "0") == "1":  # the device against the compiled reference's fixture
        nbad = 0
        for i in range(FIXTURE_N):
            try:
                test_random_case_against_reference_fixture(c, i)
            except AssertionError as e:
                nbad += 1
                print("FIXTURE CASE %d %r\n     -> %s" % (i, _fixture()[1][i], " | ".join(str(e).strip().splitlines()[:6])[:700]), flush=True)
            except BaseException as e:  # pytest.skip
                if type(e).__name__ == "Skipped":
                    continue
                if isinstance(e, (KeyboardInterrupt, SystemExit)):
                    raise
                nbad += 1
                print("FIXTURE CASE %d %r\n     -> ERROR %s: %s" % (i, _fixture()[1][i], type(e).__name__, str(e)[:500]), flush=True)
        print("%d of %d fixture cases differ from the compiled reference" % (nbad, FIXTURE_N))
        nbad = 0
        for i in range(LARGE_FIXTURE_N):
            try:
                test_random_large_case_against_reference_fixture(c, i)
            except AssertionError as e:
                nbad += 1
                print("LARGE FIXTURE CASE %d %r\n     -> %s" % (i, _fixture(True)[1][i], " | ".join(str(e).strip().splitlines()[:6])[:700]), flush=True)
            except BaseException as e:  # pytest.skip
                if type(e).__name__ == "Skipped":
                    continue
                if isinstance(e, (KeyboardInterrupt, SystemExit)):
                    raise
                nbad += 1
                print("LARGE FIXTURE CASE %d %r\n     -> ERROR %s: %s" % (i, _fixture(True)[1][i], type(e).__name__, str(e)[:500]), flush=True)
        print("%d of %d large fixture cases differ from the compiled reference" % (nbad, LARGE_FIXTURE_N))
    nbad = 0
    for i in range(NLARGE):
        try:
            test_random_large_case_against_oracle(c, i)
        except AssertionError as e:
            nbad += 1
            print("LARGE CASE %d %r\n     -> %s" % (i, large_cases()[i], " | ".join(str(e).strip().splitlines()[:6])[:700]), flush=True)
        except BaseException as e:  # pytest.skip
            if type(e).__name__ == "Skipped":
                continue
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                raise
            nbad += 1
            print("LARGE CASE %d %r\n     -> ERROR %s: %s" % (i, large_cases()[i], type(e).__name__, str(e)[:500]), flush=True)
    print("%d of %d large cases differ" % (nbad, NLARGE))
    nbad = nrun = 0
    for i in range(0, NCASES, 3):
        try:
            test_random_case_through_host_callbacks(c, i)
            nrun += 1
        except AssertionError as e:
            nbad += 1
            nrun += 1
            print("HOST CASE %d %r\n     -> %s" % (i, cases()[i], " | ".join(str(e).strip().splitlines()[:6])[:700]), flush=True)
        except BaseException as e:  # pytest.skip
            if type(e).__name__ == "Skipped":
                continue
            if isinstance(e, (KeyboardInterrupt, SystemExit)):
                raise
            nbad += 1
            nrun += 1
            print("HOST CASE %d %r\n     -> ERROR %s: %s" % (i, cases()[i], type(e).__name__, str(e)[:500]), flush=True)
    print("%d of %d host-callback cases differ" % (nbad, nrun))
    nbad = 0
    fc = facade_cases()
    for i in fc:
        try:
            test_random_case_user_library_problem_matches_builtin(c, i)
        except AssertionError as e:
            nbad += 1
            print("FACADE CASE %d %r\n     -> %s" % (i, cases()[i], " | ".join(str(e).strip().splitlines()[:6])[:700]), flush=True)
        except Exception as e:  # noqa: BLE001 - an error of the library is a finding too
            nbad += 1
            print("FACADE CASE %d %r\n     -> ERROR %s: %s" % (i, cases()[i], type(e).__name__, str(e)[:500]), flush=True)
    print("%d of %d user-library cases differ" % (nbad, len(fc)))
    nbad = 0
    for i in range(NQN):
        try:
            test_random_quasi_newton_sequence_against_oracle(c, i)
        except AssertionError as e:
            nbad += 1
            print("QN CASE %d %r\n     -> %s" % (i, qn_cases()[i], " | ".join(str(e).strip().splitlines()[:6])[:600]), flush=True)
        except Exception as e:  # noqa: BLE001
            nbad += 1
            print("QN CASE %d %r\n     -> ERROR %s: %s" % (i, qn_cases()[i], type(e).__name__, str(e)[:500]), flush=True)
    print("%d of %d quasi-Newton cases differ" % (nbad, NQN))
