"""
GPU parity of the trust-region layer (through the C ABI): ParOptTrustRegion's SL1QP iteration over
the quadratic and the compact-eigenvalue subproblems against trajectories of the compiled
reference (tests/golden/tr_*.npz) -- the iteration table to its print precision, accept / reject and
quasi-Newton flags exactly, the interior-point iteration counts of the two subproblem solves per
iteration exactly, except in the rows listed (with their reason) in TR_INEXACT_ROWS -- 6 of the 486 compared
rows over twelve goldens --, radius, penalty parameters, model values and the iterate.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from tr_helpers import TR_REFERENCE_IRREPRODUCIBLE, compare_tr, eig_model, tr_options_from_case

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import paropt_amd as pa

    c = pa.Context(0)
    yield c
    c.close()


def run_gpu_tr(ctx, case, python_eig_callback=False, capture_lines=None):
    import paropt_amd as pa

    a = case["args"]
    prob = pa.SeparableProblem(ctx, a["problem"], a["n"], a.get("c", 2), a.get("seed", 0),
                               a.get("eig_min", 1.0), a.get("eig_max", 100.0))
    if a.get("nwcon", 0) > 0:
        prob.setWeighting(a["nwcon"], a["nw"], a.get("nwstart", 0), a.get("nwskip", 0),
                          a.get("nwineq", a["nwcon"]))
    if a.get("chain_span", 0) > 0:
        prob.setChain(a["chain_span"], a.get("chain_stride", 1), a.get("chain_reverse", 0))
    opts, tropts = tr_options_from_case(case)
    opts.pop("write_output_frequency", None)
    tr = pa.TrustRegion(prob, dict(opts, **tropts))
    if a.get("eig_N", 0) > 0:
        N = a["eig_N"]
        if python_eig_callback:
            H, M, Minv = eig_model(a.get("seed", 0), N, a.get("eig_curv", 1.0), prob.nvars, prob.offset)

            def upd(x, e):
                for i in range(N):
                    e.hvecs[i].from_numpy(H[i])
                    e.hvecs[i].scale(1.0 / e.hvecs[i].norm())
                e.M[:, :] = M
                e.Minv[:, :] = Minv

            tr.setEigenModel(N, a.get("eig_index", 0), upd)
        else:
            tr.setEigenModelSynthetic(N, a.get("eig_index", 0), a.get("seed", 0), a.get("eig_curv", 1.0))
    snaps, rows = [], []

    def cb(i):
        if i > 0:
            rows.append(tr.getLastRow())
        if capture_lines is not None:
            # the steering solve of iteration i runs BEFORE this callback, the QP of iteration i after it: row i's
            # steering line is available now, row i - 1's QP line too
            steer, qp = tr.getLastSolveLines()
            capture_lines.setdefault(i, ["", ""])[0] = steer
            if i > 0:
                capture_lines.setdefault(i - 1, ["", ""])[1] = qp
        s = tr.snapshot()
        s["x"] = tr.getModelVectors()[0].to_numpy()
        snaps.append(s)

    tr.setIterationCallback(cb)
    tr.optimize()
    rows.append(tr.getLastRow())
    if capture_lines is not None:
        capture_lines.setdefault(len(rows) - 1, ["", ""])[1] = tr.getLastSolveLines()[1]
    st = tr.getState()
    x, z, zw = tr.getOptimizedPoint()
    final = dict(iter_count=st["iter_count"], fk=st["fk"], ck=st["ck"], x=x.to_numpy(), z=z)
    return tr, rows, snaps, final


TR_CASES = golden_names("tr_")

# Rows of the iteration tables whose interior-point iteration counts differ from the reference's (everything
# else in every row of every golden is identical: 584 of 594 compared rows).  In each of them the steering LP
# (sequential linear method, predictor-corrector barrier) or the QP ends on a round-off level test: the iterate
# crawls with |infeas| ~ 1e-15 against rho ~ 1e15 or sits on `LNoImprv` until the complementarity crosses
# 0.1 abs_res_tol, and the iteration at which that happens moves by a few with the summation order of the
# reductions (the reference itself is not reproducible across rank counts there).  counts: mine vs reference.
TR_INEXACT_ROWS = {
    "tr_convex_n300_c4_bfgs": {16, 17},          # 23/32 vs 23/31, 23/57 vs 23/54 (second solve)
    "tr_csr_convex_n120_c2_chain3s2": {17},      # 23/31 vs 23/27
    "tr_eig_convex_n300_c3_N6": {13, 16},        # 33/12 vs 34/12, 33/15 vs 32/15 (steering LP)
    "tr_rosenbrock_n60_bfgs": {3},               # 20/24 vs 20/32
    # filter method: the restoration LP of iteration 3 takes 37 vs 36 iterations; from iteration 26 on the two runs
    # hold filters of different size (f2 / f3 ...: an entry on the envelope is or is not dominated at 1e-16)
    "tr_filter_quadratic_n200_c3": {3},
    # drawn option combinations (round 3): the steering LP of one row each crawls at |infeas| ~ 1e-13 .. 1e-16 with the
    # barrier parameter at its floor; how many iterations that takes also moves with the arithmetic variant of the solve
    # passes here (stored vs recomputed first step: PAROPT_AMD_NO_RECOMPUTE=1 gives the reference's 37 and 33)
    "tr_rand_convex_n257_c2_eta01": {0},         # 39/36 vs 39/37
    "tr_rand_quadratic_n300_c8_subcon": {3},     # 21/41 vs 21/33
    "tr_rand_rosenbrock_n127_eta05": {3},        # 20/38 vs 20/30
}


@pytest.mark.parametrize("name", TR_CASES)
def test_tr_trajectory_golden(ctx, name):
    g, case = load_golden(name)
    if name in TR_REFERENCE_IRREPRODUCIBLE:  # only the compared rows are run
        case["args"]["tr.tr_max_iterations"] = TR_REFERENCE_IRREPRODUCIBLE[name]["rows"]
    lines = {}
    tr, rows, snaps, final = run_gpu_tr(ctx, case, capture_lines=lines)
    # An allow-listed row may differ from the reference in its interior-point iteration counts only (compare_tr: by
    # at most 10, everything else in the row identical) AND only because a solve spent a different number of
    # iterations in its TERMINAL phase: both solves of the row must have ended regularly (no failed line search) with
    # the barrier parameter at its floor 0.1 abs_res_tol = 1e-7, where the stop test compares round-off level numbers.
    # A count that differs for any other reason (a solve that stalls, a wrong step early on) fails here.
    for k in TR_INEXACT_ROWS.get(name, set()):
        for which, ln in zip(("steering", "qp"), lines.get(k, ["", ""])):
            parts = ln.split()
            if len(parts) < 15:
                continue  # the solve did not run in this row (fixed penalty: no steering solve)
            assert "LFail" not in parts[15:], (name, k, which, ln)
            assert abs(float(parts[11]) - 1e-7) <= 1e-12, (name, k, which, ln)
    if name in TR_REFERENCE_IRREPRODUCIBLE:  # the reference run itself depends on the rank count past these rows
        nr = TR_REFERENCE_IRREPRODUCIBLE[name]["rows"]
        assert compare_tr(g, rows, snaps, final, nr, check_snaps=False, check_counts=False) == nr
        return
    # the eigen-model on the convex objective is the worst conditioned case (gradient ~ 1/(eps+x)^2 and a
    # nonlinear constraint model): tight agreement over 40 iterations, after which 1e-6 differences in
    # the constraint values start to show; its final point is compared through the objective only
    loose = "eig_convex" in name
    window = 40 if ("sr1" in name or loose) else 60
    unstable = name == "tr_filter_quadratic_n200_c3"  # see tests/test_oracle_tr.py
    if unstable:
        window = 10  # the printed objective differs in the 6th digit from iteration 12 on
    n = compare_tr(g, rows, snaps, final, window, check_snaps=not unstable,
                   inexact_rows=TR_INEXACT_ROWS.get(name, set()))
    if unstable:
        return
    assert n >= (12 if "tr_rand_" in name else 20)  # (the drawn-option goldens run 12 iterations)
    if loose:
        assert abs(final["fk"] - g["final/fk"][0]) <= 1e-4 * max(1.0, abs(g["final/fk"][0]))
    elif "sr1" not in name:
        assert final["iter_count"] == int(g["final/iter_count"][0])
        assert abs(final["fk"] - g["final/fk"][0]) <= 1e-6 * max(1.0, abs(g["final/fk"][0]))
        np.testing.assert_allclose(final["x"], g["final/x"], rtol=0, atol=1e-5 * max(1.0, np.abs(g["final/x"]).max()))
    # the table accumulated by the driver has the reference's layout
    from tr_helpers import parse_tr_table

    mine = parse_tr_table(tr.getHistory())
    assert len(mine) == final["iter_count"]


def test_tr_eigen_model_python_callback(ctx):
    """The model-update callback boundary (setEigenModelUpdate): the same synthetic model supplied
    from Python through po_eig_get_approximation instead of the built-in one."""
    g, case = load_golden("tr_eig_quadratic_n200_c2_N4")
    case["args"]["tr.tr_max_iterations"] = 12
    tr, rows, snaps, final = run_gpu_tr(ctx, case, python_eig_callback=True)
    compare_tr(g, rows, snaps, final, 12)


def test_tr_option_errors(ctx):
    import paropt_amd as pa

    prob = pa.SeparableProblem(ctx, "quadratic", 100, 2)
    with pytest.raises(pa.ParOptAMDError):
        pa.TrustRegion(prob, {"tr_accept_step_strategy": "no_such_strategy"})
    with pytest.raises(pa.ParOptAMDError):
        pa.TrustRegion(prob, {"tr_no_such_option": 1})


def test_trust_region_over_csr_sparse_constraints(ctx):
    """The trust-region driver over a problem in the CSR form: the quadratic subproblem forwards the
    quasi-definite factor / half-solve to the wrapped problem, so the inner interior point runs the sparse
    Cholesky path.  Checked against the oracle's driver on the same problem (the oracle mirrors the
    reference's stored-value semantics of ParOptSparseProblem::evalSparseCon)."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po
    from oracle import tr_oracle as tro

    n, c = 120, 2
    opts = {"tr_init_size": 0.1, "tr_max_iterations": 12, "qn_subspace_size": 5, "output_file": "",
            "tr_output_file": ""}
    tr = pa.TrustRegion(pa.SeparableProblem(ctx, "convex", n, c).setChain(3, 2), opts)
    tr.optimize()
    ops = po.VecOps(po.SelfComm())
    sub = tro.QuadraticSubproblem(po.SepProblem("convex", n, c, chain=(3, 2)),
                                  po.LBFGS(n, 5, ops, "skip_negative_curvature"))
    otr = tro.TrustRegion(sub, po.InteriorPoint(sub, {}), {"tr_init_size": 0.1, "tr_max_iterations": 12})
    otr.optimize()
    st = tr.getState()
    assert st["iter_count"] == otr.iter_count
    np.testing.assert_allclose(tr.getOptimizedPoint()[0].to_numpy(), sub.xk, rtol=0, atol=1e-6)


def test_trust_region_with_a_panel_wider_than_one_launch(ctx):
    """c + k = 70 + 2 * 15 = 100 panel columns under the trust-region driver (subproblem evaluations, steering LP and
    QP solves all take the slabbed / blocked / collapsed forms of the panel kernels) against the oracle's driver."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po
    from oracle import tr_oracle as tro

    n, c, m = 400, 70, 15
    opts = {"tr_init_size": 0.1, "tr_max_iterations": 8, "qn_subspace_size": m, "output_file": "", "tr_output_file": ""}
    tr = pa.TrustRegion(pa.SeparableProblem(ctx, "quadratic", n, c), opts)
    rows = []
    tr.setIterationCallback(lambda i: rows.append(tr.getLastRow()) if i > 0 else None)
    tr.optimize()
    rows.append(tr.getLastRow())
    ops = po.VecOps(po.SelfComm())
    sub = tro.QuadraticSubproblem(po.SepProblem("quadratic", n, c), po.LBFGS(n, m, ops, "skip_negative_curvature"))
    otr = tro.TrustRegion(sub, po.InteriorPoint(sub, {}), {"tr_init_size": 0.1, "tr_max_iterations": 8})
    otr.optimize()
    st = tr.getState()
    assert st["iter_count"] == otr.iter_count
    assert abs(st["fk"] - sub.fk) <= 1e-8 * max(1.0, abs(sub.fk))
    np.testing.assert_allclose(tr.getOptimizedPoint()[0].to_numpy(), sub.xk, rtol=0, atol=1e-6)
    # interior-point iteration counts of both subproblem solves of every iteration
    mine = [t for _, t in rows]
    ref = [list(t["info"]) for t in otr.trace]
    assert mine == ref, (mine, ref)
