"""Shared helpers of the trust-region parity tests (oracle and GPU): case -> objects, table parsing."""
import numpy as np

from conftest import ip_options_from_case

COLS = ("fobj", "infeas", "l1", "linfty", "smax", "tr", "rho", "model_reduc", "zav", "zmax", "gav", "gmax")


def parse_tr_table(text):
    """{iter: ([12 numeric columns], [info tokens])} of a paropt.tr table (:1424-1438); the
    wall-time column is dropped."""
    rows = {}
    for ln in str(text).splitlines():
        p = ln.split()
        if len(p) >= 14 and p[0].isdigit():
            rows[int(p[0])] = ([float(v) for v in p[1:13]], p[14:])
    return rows


def tr_options_from_case(case):
    a = case["args"]
    opts = ip_options_from_case(case)
    tropts = {k[3:]: v for k, v in a.items() if k.startswith("tr.")}
    return opts, tropts


def eig_model(seed, N, curv, nlocal, offset=0):
    """The synthetic eigenvalue model of oracle/ref_driver.cpp eig_update: unit hash directions,
    M = -curv (1 + 0.1 i) I.  Returns (H [N, nlocal] un-normalised, M, Minv)."""
    from oracle import paropt_oracle as po

    idx = np.arange(offset, offset + nlocal, dtype=np.uint64)
    H = np.stack([2.0 * po.u01(seed, 300 + i, idx) - 1.0 for i in range(N)])
    d = np.array([-curv * (1.0 + 0.1 * i) for i in range(N)])
    return H, np.diag(d), np.diag(1.0 / d)


def run_oracle_tr(case, nmax=None):
    from oracle import paropt_oracle as po
    from oracle import tr_oracle as tro

    a = case["args"]
    prob = po.SepProblem(a["problem"], a["n"], a.get("c", 2), seed=a.get("seed", 0),
                         eig_min=a.get("eig_min", 1.0), eig_max=a.get("eig_max", 100.0),
                         nwcon=a.get("nwcon", 0), nw=a.get("nw", 0), nwstart=a.get("nwstart", 0),
                         nwskip=a.get("nwskip", 0), nwineq=a.get("nwineq", -1),
                         chain=(a["chain_span"], a.get("chain_stride", 1)) if a.get("chain_span", 0) else None)
    opts, tropts = tr_options_from_case(case)
    if "penalty_gamma" in opts:
        tropts["penalty_gamma"] = opts["penalty_gamma"]
    if nmax:
        tropts["tr_max_iterations"] = nmax
    ops = po.VecOps(prob.comm)
    qt, m = opts.get("qn_type", "bfgs"), opts.get("qn_subspace_size", 10)
    if qt == "bfgs":
        qn = po.LBFGS(prob.nlocal, m, ops, opts.get("qn_update_type", "skip_negative_curvature"))
    else:
        qn = po.LSR1(prob.nlocal, m, ops)
    if a.get("eig_N", 0) > 0:
        N = a["eig_N"]
        H, M, Minv = eig_model(a.get("seed", 0), N, a.get("eig_curv", 1.0), prob.nlocal)
        eigh = tro.CompactEigenApprox(prob.nlocal, N, ops)

        def upd(x, e):
            for i in range(N):
                e.hvecs[i] = H[i] / ops.norm(H[i])
            e.M[:, :] = M
            e.Minv[:, :] = Minv

        sub = tro.EigenSubproblem(prob, tro.EigenQuasiNewton(qn, eigh, a.get("eig_index", 0)), upd)
    else:
        sub = tro.QuadraticSubproblem(prob, qn)
    ip = po.InteriorPoint(sub, opts)
    tr = tro.TrustRegion(sub, ip, tropts)
    snaps = []

    def hook(t, i):
        b0, d0, M_, Z = sub.get_quasi_newton().get_compact()
        snaps.append(dict(tr_size=t.tr_size, penalty_gamma=t.penalty_gamma.copy(), fk=sub.fk, ck=sub.ck.copy(),
                          iters=np.array([t.iter_count, t.subproblem_iters, t.adaptive_subproblem_iters]),
                          norms=np.array([ops.norm(sub.xk), ops.norm(sub.gk)]), qn_size=len(Z), qn_b0=b0,
                          x=sub.xk.copy()))

    tr.hook = hook
    tr.optimize()
    rows = [([t[c] for c in COLS], t["info"]) for t in tr.trace]
    final = dict(iter_count=tr.iter_count, fk=sub.fk, ck=sub.ck.copy(), x=sub.xk.copy(), tr_size=tr.tr_size,
                 penalty_gamma=tr.penalty_gamma.copy(), z=ip.vars.z.copy())
    return rows, snaps, final


# Goldens whose REFERENCE run is not reproducible across MPI rank counts beyond the stated number of rows, so that
# no implementation can agree with them further.  tr_convex_n100000_c32_sr1_r4 is the metric's configuration (config
# 3 shape: convex objective, c = 32, L-SR1(10)) under the trust-region driver at n = 1e5: the steering LP of
# iteration 3 crawls into its 200-iteration cap, and the unmodified reference run on 1 / 2 / 4 ranks prints
# fobj = 8.70061e+05 / 8.70805e+05 / 8.69399e+05 at iteration 4 and interior-point counts 26/74, 26/73, 26/47 at
# iteration 1 (oracle/make_golden.py; recorded here on 4 ranks).  Compared: the first `rows` rows of the table to
# print precision and the accept / reject flags; not the interior-point counts, not the snapshots past the window.
TR_REFERENCE_IRREPRODUCIBLE = {"tr_convex_n100000_c32_sr1_r4": dict(rows=4)}


def compare_tr(g, rows, snaps, final, window, frac_exact=0.8, check_snaps=True, inexact_rows=None,
               check_counts=True):
    """rows/snaps/final of a run (oracle or device) against a golden of the compiled reference.
    inexact_rows (a set of iteration numbers): the info strings of all OTHER rows -- accept / reject and
    quasi-Newton flags, interior-point iteration counts of both subproblem solves -- must be identical; in the
    listed rows only the counts may differ (by at most 10 iterations)."""
    ref = parse_tr_table(g["paropt_tr"])
    ncmp = min(window, len(ref), len(rows))
    assert ncmp >= min(window, len(ref)), (ncmp, len(ref), len(rows))
    exact = 0
    for k in range(ncmp):
        vals, toks = rows[k]
        rvals, rtoks = ref[k]
        for name, a, b in zip(COLS, vals, rvals):
            # the table prints 3-6 significant digits; quantities that are differences of nearly
            # equal numbers (rho, model reduction at convergence) get an absolute floor
            tol = 6e-3 * abs(b) + (1e-9 if name in ("infeas", "smax", "model_reduc") else 1e-12)
            if name == "rho":
                tol = 2e-2 * max(1.0, abs(b))
            if name == "fobj":
                tol = 2e-5 * max(1e-3, abs(b))
            assert abs(a - b) <= tol, "%s @%d: %r vs %r" % (name, k, a, b)
        flags = [t for t in toks if "/" not in t and not t.isdigit()]
        rflags = [t for t in rtoks if "/" not in t and not t.isdigit()]
        assert flags == rflags, "flags @%d: %s vs %s" % (k, toks, rtoks)
        exact += int(toks == rtoks)
        if not check_counts:
            continue
        if inexact_rows is not None and toks != rtoks:
            assert k in inexact_rows, "info @%d: %s vs %s" % (k, toks, rtoks)
            mine_n = [int(v) for t in toks if "/" in t or t.isdigit() for v in t.split("/")]
            ref_n = [int(v) for t in rtoks if "/" in t or t.isdigit() for v in t.split("/")]
            assert len(mine_n) == len(ref_n) and max(abs(a - b) for a, b in zip(mine_n, ref_n)) <= 10, (k, toks, rtoks)
    # interior-point iteration counts of the two subproblem solves: bit-exact except where the
    # degenerate steering LP terminates on a round-off level test (see DESIGN.md "Parity")
    assert exact >= frac_exact * ncmp or not check_counts, "only %d of %d info strings identical" % (exact, ncmp)
    for k in range(min(ncmp, len(snaps)) if check_snaps else 0):
        p = "tr%03d/" % k
        s = snaps[k]
        assert abs(s["tr_size"] - g[p + "tr_size"][0]) <= 1e-12 * g[p + "tr_size"][0], k
        np.testing.assert_allclose(s["penalty_gamma"], g[p + "penalty_gamma"], rtol=1e-5, err_msg="gamma @%d" % k)
        assert abs(s["fk"] - g[p + "fk"][0]) <= 1e-6 * max(1.0, abs(g[p + "fk"][0])), k
        np.testing.assert_allclose(s["ck"], g[p + "ck"], rtol=1e-6,
                                   atol=1e-6 * max(1.0, np.abs(g[p + "ck"]).max()), err_msg="ck @%d" % k)
        assert int(s["iters"][0]) == int(g[p + "iters"][0])
        # |xk| tightly; |gk| looser: the convex objective's gradient -b^2/(eps+x)^2 amplifies 1e-7
        # differences in x by 1e3 near the lower bound
        np.testing.assert_allclose(s["norms"][0], g[p + "norms"][0], rtol=1e-6, err_msg="|xk| @%d" % k)
        np.testing.assert_allclose(s["norms"][1], g[p + "norms"][1], rtol=2e-4, err_msg="|gk| @%d" % k)
        assert s["qn_size"] == int(g[p + "qn_size"][0]), "qn size @%d" % k
        if p + "x" in g and "x" in s:
            np.testing.assert_allclose(s["x"], g[p + "x"], rtol=0, atol=1e-6 * max(1.0, np.abs(g[p + "x"]).max()),
                                       err_msg="x @%d" % k)
    return ncmp
