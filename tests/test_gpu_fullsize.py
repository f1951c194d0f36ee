"""
Size-independent properties at BASELINE.json's full sizes (where the numpy oracle is too slow to
be the checker): linearity of mdot, consistency of the MFMA weighted Gram with mdot, run-to-run
bitwise determinism, interior-point invariants (strict interiority, positive multipliers,
non-increasing barrier parameter), and oracle parity at n = 1e6.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_METRIC = 50_000_000  # config 3: n = 50 M, m = 32
N_C2 = 10_000_000      # config 2: n = 10 M, m = 8, L-BFGS(20)


@pytest.fixture(scope="module")
def ctx():
    import paropt_amd as pa

    c = pa.Context(0)
    yield c
    c.close()


def test_mdot_linearity_and_wgram_consistency_n50M(ctx):
    import paropt_amd as pa

    n, nv = N_METRIC, 32
    x = pa.PVec(ctx, n).fill_hash(0, 10, 0, 2.0, -1.0)
    y = pa.PVec(ctx, n).fill_hash(0, 11, 0, 2.0, -1.0)
    V = [pa.PVec(ctx, n).fill_hash(0, 20 + j, 0, 2.0, -1.0) for j in range(nv)]
    mx, my = x.mdot(V), y.mdot(V)
    # E[x v] = 0, var = n/9: |dot| ~ sqrt(n)/3 ~ 2.4e3; rounding bound 4 eps sqrt(n) sum|terms| ~ 1e-6
    assert np.all(np.abs(mx) < 6 * np.sqrt(n) / 3)
    x.axpy(-0.375, y)
    np.testing.assert_allclose(x.mdot(V), mx - 0.375 * my, rtol=0, atol=2e-6)
    # run-to-run determinism: bitwise
    np.testing.assert_array_equal(y.mdot(V), my)
    # weighted Gram vs mdot on the same data: W[:, j] = mdot(d * V_j, V)
    d = pa.PVec(ctx, n).fill_hash(0, 9, 0, 1.0, 0.5)
    W = pa.wgram(d, V)
    np.testing.assert_array_equal(W, W.T)
    assert np.all(np.linalg.eigvalsh(W) > 0)  # P^T D P with D > 0 and n >> nv is SPD
    ones = pa.PVec(ctx, n)
    ones.set(1.0)
    W1 = pa.wgram(ones, V)
    for j in (0, 13, 31):
        np.testing.assert_allclose(W1[:, j], V[j].mdot(V), rtol=0, atol=2e-6 + 1e-12 * n)
    np.testing.assert_array_equal(pa.wgram(d, V), W)


@pytest.mark.parametrize("problem,n,c,qn,m,iters", [
    ("convex", N_METRIC, 32, "sr1", 10, 14),     # the metric's configuration (config 3)
    ("quadratic", N_C2, 8, "bfgs", 20, 25),      # config 2
])
def test_ip_invariants_fullsize(ctx, problem, n, c, qn, m, iters):
    import paropt_amd as pa

    opts = {"qn_type": qn, "qn_subspace_size": m, "abs_res_tol": 1e-30, "start_affine_multiplier_min": 0.01,
            "max_major_iters": iters, "write_output_frequency": 0}

    def run():
        ip = pa.InteriorPoint(pa.SeparableProblem(ctx, problem, n, c), opts)
        sn = []
        ip.setIterationCallback(lambda k: sn.append(ip.snapshot()))
        ip.optimize()
        return ip, sn

    ip, sn = run()
    assert ip.getIterationCounters()[0] == iters
    lo, hi = (0.0, 1.0) if problem == "convex" else (-5.0, 5.0)
    x, z, zl, zu = ip.getOptimizedPoint()
    xa = x.to_numpy()
    assert xa.min() > lo and xa.max() < hi                      # strict interiority (clamps :3156-3190)
    assert zl.to_numpy().min() > 0.0 and zu.to_numpy().min() > 0.0
    s, t, zs, zt = ip.getOptimizedSlacks()
    assert min(s.min(), t.min(), zs.min(), zt.min()) > 0.0
    mus = [q["mu"] for q in sn]
    assert all(b <= a * (1 + 1e-12) for a, b in zip(mus[1:], mus[2:]))   # monotone barrier
    for a, b in zip(sn, sn[1:]):
        assert b["counters"][0] == a["counters"][0] + 1
        assert b["counters"][1] >= a["counters"][1] + 1         # >= one evaluation per iteration
        assert b["counters"][2] in (a["counters"][2], a["counters"][2] + 1)
        assert np.all(np.isfinite(b["norms"])) and np.isfinite(b["fobj"])
    if qn == "bfgs":
        assert sn[-1]["fobj"] < sn[1]["fobj"]                     # convergent variant makes progress
    # bitwise run-to-run determinism of the whole trajectory
    ip2, sn2 = run()
    for a, b in zip(sn, sn2):
        assert a["fobj"] == b["fobj"] and a["mu"] == b["mu"]
        np.testing.assert_array_equal(a["norms"], b["norms"])
        np.testing.assert_array_equal(a["z"], b["z"])


def test_config4_fullsize_weighting(ctx):
    """BASELINE config 4 at full size on one GPU: n = 20 M, 4 dense + 1 M weighting constraints (groups of 20),
    L-BFGS(10): invariants of the sparse blocks, feasibility of the weighting constraints along the run and
    bitwise run-to-run determinism."""
    import paropt_amd as pa

    n, nwcon, nw, iters = 20_000_000, 1_000_000, 20, 12
    opts = {"qn_type": "bfgs", "qn_subspace_size": 10, "abs_res_tol": 1e-30, "start_affine_multiplier_min": 0.01,
            "max_major_iters": iters, "write_output_frequency": 0}

    def run():
        prob = pa.SeparableProblem(ctx, "convex", n, 4)
        prob.setWeighting(nwcon, nw, 0, 0)
        ip = pa.InteriorPoint(prob, opts)
        sn = []
        ip.setIterationCallback(lambda k: sn.append(ip.snapshot()))
        ip.optimize()
        return ip, sn

    ip, sn = run()
    assert ip.getIterationCounters()[0] == iters
    x = ip.getOptimizedPoint()[0].to_numpy()
    assert x.min() > 0.0 and x.max() < 1.0
    zw, sw, tw, zsw, ztw = (v.to_numpy() for v in ip.getOptimizedSparse())
    assert min(sw.min(), tw.min(), zsw.min(), ztw.min()) > 0.0
    assert zw.shape == (nwcon,)
    # infeasibility (the |infes| column: dense and weighting constraints in slack form) does not grow over the run
    rows = [ln.split() for ln in ip.getHistory().splitlines() if ln[:5].strip().isdigit()]
    assert len(rows) == iters and float(rows[-1][9]) <= float(rows[0][9])
    for a, b in zip(sn, sn[1:]):
        assert b["counters"][0] == a["counters"][0] + 1
        assert np.all(np.isfinite(b["norms"])) and np.all(np.isfinite(b["wnorms"])) and np.isfinite(b["fobj"])
    ip2, sn2 = run()
    for a, b in zip(sn, sn2):
        assert a["fobj"] == b["fobj"] and a["mu"] == b["mu"]
        np.testing.assert_array_equal(a["wnorms"], b["wnorms"])


def test_config5_fullsize_trust_region_eigen(ctx):
    """BASELINE config 5 at full size: the trust-region driver over the compact-eigenvalue subproblem at
    n = 5 M (4 constraints, N = 10 curvature directions, L-BFGS(10)): radius within its limits, finite table,
    monotone acceptance bookkeeping, and bitwise determinism of the whole run."""
    import paropt_amd as pa

    n = 5_000_000

    def run():
        tr = pa.TrustRegion(pa.SeparableProblem(ctx, "quadratic", n, 4),
                            {"tr_max_iterations": 5, "qn_subspace_size": 10, "max_major_iters": 200,
                             "tr_output_file": "", "output_file": ""})
        tr.setEigenModelSynthetic(10, 0, 0, 2.0)
        sn = []
        tr.setIterationCallback(lambda k: sn.append(tr.snapshot()))
        tr.optimize()
        return tr, sn

    tr, sn = run()
    assert len(sn) >= 5
    for k, s in enumerate(sn):
        assert 1e-3 <= s["tr_size"] <= 1.0 + 1e-12       # tr_min_size <= radius <= tr_max_size
        assert np.isfinite(s["fk"]) and np.all(np.isfinite(s["ck"])) and np.all(np.isfinite(s["norms"]))
        assert k == 0 or s["iters"][1] > 0                # every iteration solved a subproblem
    x = tr.getOptimizedPoint()[0].to_numpy()
    assert x.min() >= -5.0 and x.max() <= 5.0
    tr2, sn2 = run()
    for a, b in zip(sn, sn2):
        assert a["fk"] == b["fk"] and a["tr_size"] == b["tr_size"]
        np.testing.assert_array_equal(a["iters"], b["iters"])


def test_ip_vs_oracle_n1M(ctx):
    """Oracle parity at the largest size the numpy oracle finishes in about a minute."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    n, c = 1_000_003, 8
    opts = {"qn_subspace_size": 10, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01, "max_major_iters": 10}
    oip = po.InteriorPoint(po.SepProblem("quadratic", n, c), opts)
    osn = []
    oip.hook = lambda s, k: osn.append(s.snapshot())
    oip.optimize()
    ip = pa.InteriorPoint(pa.SeparableProblem(ctx, "quadratic", n, c), dict(opts, write_output_frequency=0))
    gsn = []
    ip.setIterationCallback(lambda k: gsn.append(ip.snapshot()))
    ip.optimize()
    assert len(gsn) == len(osn)
    for g, o in zip(gsn, osn):
        np.testing.assert_array_equal(g["counters"], o["counters"])
        assert g["qn_size"] == o["qn_size"]
        assert abs(g["fobj"] - o["fobj"]) <= 1e-7 * max(1.0, abs(o["fobj"]))
        np.testing.assert_allclose(g["norms"], o["norms"], rtol=1e-7)
        np.testing.assert_allclose(g["z"], o["z"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(ip.getOptimizedPoint()[0].to_numpy(), oip.vars.x, rtol=0, atol=1e-7)
