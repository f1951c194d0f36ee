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
