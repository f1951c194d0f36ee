"""
The metric's workload through the REAL drop-in boundary: examples/random_convex_amd.cpp is a ParOptProblem subclass
on include/ParOptAMD.hpp (the reference's interface, src/ParOptProblem.h:42-296; the reference's instance of the
problem: examples/random_convex/random_convex.py:44-126) with the user's own HIP kernels, compiled outside
libparopt_amd.so.  It must drive the solver exactly like the library's built-in twin of the same problem:
integer bookkeeping bit-exact, state to round-off (the two differ only in the summation order of the objective).
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
USER_LIB = os.path.join(ROOT, "examples", "librandom_convex_user.so")


@pytest.fixture(scope="module")
def ctx():
    import paropt_amd as pa

    c = pa.Context(0)
    yield c
    c.close()


def tokens(text):
    out = {}
    for ln in str(text).splitlines():
        parts = ln.split()
        if len(parts) >= 15 and parts[0].isdigit():
            out[int(parts[0])] = parts[15:]
    return out


def run(ctx, prob, qn, iters, **extra):
    import paropt_amd as pa

    opts = {"qn_type": qn, "qn_subspace_size": 10, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01,
            "max_major_iters": iters, "write_output_frequency": 0}
    opts.update(extra)
    ip = pa.InteriorPoint(prob, opts)
    snaps = []
    ip.setIterationCallback(lambda k: snaps.append(ip.snapshot()))
    red0, lau0 = ctx.counters()
    ip.optimize()
    red1, lau1 = ctx.counters()
    x, z, zl, zu = ip.getOptimizedPoint()
    return dict(ip=ip, snaps=snaps, x=x.to_numpy(), z=np.array(z), counters=ip.getIterationCounters(),
                hist=ip.getHistory(), syncs=red1 - red0, launches=lau1 - lau0)


@pytest.mark.parametrize("qn,n,c,iters", [("bfgs", 20011, 5, 40), ("bfgs", 100000, 32, 30), ("sr1", 100000, 32, 12)])
def test_user_problem_on_the_facade_matches_the_builtin(ctx, qn, n, c, iters):
    import paropt_amd as pa

    assert os.path.exists(USER_LIB), "examples/librandom_convex_user.so is not built (__graft_entry__.build())"
    a = run(ctx, pa.SeparableProblem(ctx, "convex", n, c), qn, iters)
    user = pa.UserLibraryProblem(ctx, USER_LIB, n, c)
    b = run(ctx, user, qn, iters)
    assert a["counters"] == b["counters"]
    ta, tb = tokens(a["hist"]), tokens(b["hist"])
    assert ta == tb
    for sa, sb in zip(a["snaps"], b["snaps"]):
        np.testing.assert_array_equal(sa["counters"], sb["counters"])
        assert sa.get("qn_size", 0) == sb.get("qn_size", 0)
        for key in ("gpiv", "mfpiv", "clamped"):
            if key in sa:
                np.testing.assert_array_equal(np.asarray(sa[key]), np.asarray(sb[key]), err_msg=key)
        assert abs(sa["mu"] - sb["mu"]) <= 1e-9 * abs(sa["mu"])
        assert abs(sa["fobj"] - sb["fobj"]) <= 1e-10 * max(1.0, abs(sa["fobj"]))
    np.testing.assert_allclose(b["x"], a["x"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(b["z"], a["z"], rtol=1e-6, atol=1e-6 * max(1.0, np.abs(a["z"]).max()))
    # deferred reductions (opt-in extension): the same bits as the immediate form, fewer host synchronisations
    user2 = pa.UserLibraryProblem(ctx, USER_LIB, n, c).setDeferredReductions(True)
    d = run(ctx, user2, qn, iters)
    assert d["counters"] == b["counters"]
    np.testing.assert_array_equal(d["x"], b["x"])
    np.testing.assert_array_equal(d["z"], b["z"])
    assert d["syncs"] < b["syncs"]
    user.close()
    user2.close()


def test_user_problem_evaluations_against_numpy(ctx):
    """The user kernels themselves: f, c, g and the rewritten Jacobian against the formulas in numpy."""
    import paropt_amd as pa
    from oracle import paropt_oracle as po
    import ctypes as C
    from paropt_amd import lib as L

    n, c = 4097, 3
    user = pa.UserLibraryProblem(ctx, USER_LIB, n, c)
    idx = np.arange(n, dtype=np.uint64)
    b = po.u01(0, 2, idx)
    A = np.stack([po.u01(0, 100 + j, idx) for j in range(c)])
    x = pa.PVec(ctx, n).fill_hash(0, 3, 0, 0.9, 0.05)
    xn = 0.05 + 0.9 * po.u01(0, 3, idx)
    f = C.c_double()
    con = np.zeros(c)
    assert L.lib.po_problem_eval_obj_con(user.handle, x.handle, C.byref(f), con.ctypes.data_as(L.c_double_p)) == 0
    np.testing.assert_allclose(f.value, np.sum(b * b / (1e-3 + xn)), rtol=1e-13)
    np.testing.assert_allclose(con, 0.25 * A.sum(axis=1) - A @ xn, rtol=0, atol=1e-10)
    g = pa.PVec(ctx, n)
    Ac = [pa.PVec(ctx, n) for _ in range(c)]
    arr = (L.po_vec * c)(*[v.handle for v in Ac])
    assert L.lib.po_problem_eval_obj_con_gradient(user.handle, x.handle, g.handle, arr) == 0
    np.testing.assert_allclose(g.to_numpy(), -(b * b) / (1e-3 + xn) ** 2, rtol=1e-14)
    for j in range(c):
        np.testing.assert_array_equal(Ac[j].to_numpy(), -A[j])
    user.close()


# ---- config 4 through the exported boundary (round 5) -----------------------------------------------------------------
WEIGHTING_LIB = os.path.join(ROOT, "examples", "libweighting_user.so")


def sparse_factor_info(prob):
    import paropt_amd.lib as L

    L.lib.po_quasidef_factor_info.restype = __import__("ctypes").c_char_p
    t = L.lib.po_quasidef_factor_info(prob.handle)
    return t.decode() if t else ""


@pytest.mark.parametrize("n,c,nw,iters", [(20000, 4, 20, 40), (100000, 4, 20, 30), (4096, 3, 8, 30)])
def test_user_weighting_problem_takes_the_fused_group_path(ctx, n, c, nw, iters):
    """examples/weighting_amd.cpp: config 4's workload as a user's ParOptSparseProblem (src/ParOptProblem.h:301-335) --
    the weighting constraints reach the library ONLY as the CSR pattern of setSparseJacobianData plus the entries its
    gradient callback writes.  The library recognises the grouped pattern and must then run exactly the kernels the
    built-in twin runs: same number of launches and host synchronisations per optimize(), integer bookkeeping bit-exact,
    state to round-off (the user's kernels sum f and cw in another order)."""
    import paropt_amd as pa

    assert os.path.exists(WEIGHTING_LIB), "examples/libweighting_user.so is not built (__graft_entry__.build())"
    nwcon = n // nw
    a = run(ctx, pa.SeparableProblem(ctx, "convex", n, c).setWeighting(nwcon, nw, 0, 0), "bfgs", iters)
    user = pa.UserLibraryProblem(ctx, WEIGHTING_LIB, n, c, prefix="wt", nwcon=nwcon, nw=nw)
    b = run(ctx, user, "bfgs", iters)
    assert sparse_factor_info(user).startswith("nblock: 1"), sparse_factor_info(user)  # the scalar block form
    assert a["counters"] == b["counters"]
    assert tokens(a["hist"]) == tokens(b["hist"])
    for sa, sb in zip(a["snaps"], b["snaps"]):
        np.testing.assert_array_equal(sa["counters"], sb["counters"])
        assert sa.get("qn_size", 0) == sb.get("qn_size", 0)
        for key in ("gpiv", "mfpiv", "clamped"):
            if key in sa:
                np.testing.assert_array_equal(np.asarray(sa[key]), np.asarray(sb[key]), err_msg=key)
        # (two device runs whose f and cw are summed in another order: round-off level differences grow along the
        # trajectory like any reduction-order change, cf. tests/golden/self_disagreement.json)
        assert abs(sa["mu"] - sb["mu"]) <= 1e-7 * abs(sa["mu"])
        assert abs(sa["fobj"] - sb["fobj"]) <= 1e-7 * max(1.0, abs(sa["fobj"]))
        np.testing.assert_allclose(sb["wnorms"], sa["wnorms"], rtol=1e-6)
    np.testing.assert_allclose(b["x"], a["x"], rtol=0, atol=1e-6)
    # the library's launches: the built-in problem's evaluation kernels are the library's own and counted, the user's
    # are not -- per optimize() the user route shows FEWER library launches, and never the column-by-column path
    # (that would be ~ (c + k) launches more per iteration)
    # (measured: 4.4 more per iteration -- the entry check and its final stage, the callbacks' own ParOptVec::mdot with its
    # final stage where the built-in problem batches; the column-by-column path would add c + k + 1 >= 25)
    assert b["launches"] <= a["launches"] + 8 * iters, (a["launches"], b["launches"])
    user.close()


def _group_pattern(nwcon, nw, start=0, skip=0):
    rowp = np.arange(nwcon + 1, dtype=np.intc) * nw
    cols = (start + (np.arange(nwcon)[:, None] * (nw + skip)) + np.arange(nw)[None, :]).astype(np.intc).ravel()
    return rowp, cols


class _PyWeighting:
    """The golden problems with weighting constraints (oracle SepProblem) handed over as a Python CSR problem: the
    reference's ParOptSparseProblem interface in its ctypes form (paropt_amd.Problem with rowp / cols)."""

    def __new__(cls, ctx, oprob, nwineq, scale=None):
        import paropt_amd as pa

        rowp, cols = _group_pattern(oprob.nwcon, oprob.nw, oprob.nwstart, oprob.nwskip)

        class P(pa.Problem):
            def getVarsAndBounds(self, x, lb, ub):
                x0, l0, u0 = oprob.vars_and_bounds()
                x[:], lb[:], ub[:] = x0, l0, u0

            def evalSparseObjCon(self, x, sparse):
                fail, f, con = oprob.eval_obj_con(x)
                sparse[:] = oprob.eval_sparse_con(x)
                return fail, f, con

            def evalSparseObjConGradient(self, x, g, A, data):
                _, gg, aa = oprob.eval_obj_con_gradient(x)
                g[:] = gg
                for j in range(oprob.c):
                    A[j][:] = aa[j]
                data[:] = -1.0 if scale is None else scale(len(data))
                return 0

        return P(ctx, oprob.nlocal, oprob.c, oprob.c, nwcon=oprob.nwcon, nwinequality=nwineq, rowp=rowp, cols=cols)


IPW_VIA_CSR = ["ipw_convex_n400_c4_w80", "ipw_convex_n300_c2_w30_eq", "ipw_rosenbrock_n100_w5",
               "ipw_convex_n240_c3_w40_l2", "ipw_convex_n240_c3_w40_mehrotra"]


@pytest.mark.parametrize("name", IPW_VIA_CSR)
def test_weighting_goldens_through_the_csr_interface(ctx, name):
    """The reference-run goldens with weighting constraints, the constraints handed over as a CSR pattern (Python
    ParOptSparseProblem form): the recognised pattern takes the block path and follows the reference's trajectory --
    counters and info tokens exactly, state within the tolerance schedule of tests/conftest.py."""
    import paropt_amd as pa
    from conftest import ip_options_from_case, load_golden, tolerance_schedule
    from oracle import paropt_oracle as po

    g, case = load_golden(name)
    a = case["args"]
    oprob = po.SepProblem(a["problem"], a["n"], a.get("c", 2), nwcon=a["nwcon"], nw=a["nw"],
                          nwstart=a.get("nwstart", 0), nwskip=a.get("nwskip", 0), nwineq=a.get("nwineq", a["nwcon"]))
    prob = _PyWeighting(ctx, oprob, a.get("nwineq", a["nwcon"]))
    opts = ip_options_from_case(case)
    opts["write_output_frequency"] = 0
    ip = pa.InteriorPoint(prob, opts)
    snaps = []
    ip.setIterationCallback(lambda k: snaps.append(ip.snapshot()))
    ip.optimize()
    assert sparse_factor_info(prob).startswith("nblock: 1")
    nref = 1 + max(int(k[2:5]) for k in g if k.startswith("it") and k.endswith("/mu"))
    ncmp = min(nref, len(snaps))
    assert ncmp >= nref - 1
    tol = tolerance_schedule(name)
    for k in range(ncmp):
        p = "it%03d/" % k
        s = snaps[k]
        np.testing.assert_array_equal(s["counters"], g[p + "counters"], err_msg="counters @%d" % k)
        assert abs(s["mu"] - g[p + "mu"][0]) / abs(g[p + "mu"][0]) <= tol("mu", k), k
        assert abs(s["fobj"] - g[p + "fobj"][0]) / max(1.0, abs(g[p + "fobj"][0])) <= tol("fobj", k), k
        for key in ("z", "s", "t", "zs", "zt"):
            ref = g[p + key]
            if ref.size:
                assert np.abs(s[key] - ref).max() / max(1.0, np.abs(ref).max()) <= tol("dense", k), (key, k)
        wa, wb = np.asarray(s["wnorms"]), np.asarray(g[p + "wnorms"])
        nz = wb != 0
        assert (np.abs(wa[nz] - wb[nz]) / np.abs(wb[nz])).max() <= tol("wnorms", k), k


def test_grouped_pattern_with_nonuniform_entries_falls_back_to_the_general_csr_path(ctx):
    """Same pattern, entries that are NOT all equal (weights 1, 2, 3, ... inside a group): the recognition must be given
    up at the first gradient evaluation (device check of the entries) and the general CSR path -- analysis + sparse
    Cholesky -- must solve the problem; compared with the same problem when the recognition is switched off."""
    import subprocess
    import sys

    code = r'''
import numpy as np, sys, json
sys.path.insert(0, %r); sys.path.insert(0, %r)
import paropt_amd as pa
from oracle import paropt_oracle as po
from test_gpu_user_problem import _PyWeighting, sparse_factor_info
ctx = pa.Context(0)
oprob = po.SepProblem("convex", 240, 3, nwcon=40, nw=5, nwstart=0, nwskip=1, nwineq=40)
w = np.tile(-(1.0 + 0.25 * np.arange(5)), 40)  # the Jacobian entries, row by row
oprob.eval_sparse_con = lambda x: 1.0 + (w.reshape(40, 5) * x[:240].reshape(40, 6)[:, :5]).sum(axis=1)
prob = _PyWeighting(ctx, oprob, 40, scale=lambda n: w)
ip = pa.InteriorPoint(prob, {"qn_subspace_size": 5, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01,
                             "max_major_iters": 25, "write_output_frequency": 0})
ip.optimize()
x = ip.getOptimizedPoint()[0].to_numpy()
print(json.dumps({"counters": ip.getIterationCounters(), "x": x.tolist(), "info": sparse_factor_info(prob)}))
''' % (ROOT, os.path.join(ROOT, "tests"))
    outs = []
    for env_extra in ({}, {"PAROPT_AMD_NO_CSR_GROUPS": "1"}):
        env = dict(os.environ, **env_extra)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((__import__("json").loads(r.stdout.strip().splitlines()[-1]), r.stderr))
    (a, erra), (b, errb) = outs
    assert "general CSR path" in erra and "general CSR path" not in errb
    assert not a["info"].startswith("nblock") and not b["info"].startswith("nblock")
    assert a["counters"] == b["counters"]
    np.testing.assert_allclose(a["x"], b["x"], rtol=0, atol=1e-9)
