"""
GPU parity of the compact quasi-Newton classes (C ABI) against the golden sequences of the
compiled reference: update return codes and compact-matrix size bit-exact; b0, d0, M to 1e-11;
mult / multAdd to 1e-8 (L-BFGS) and 5e-6 (L-SR1, ill-conditioned M with the 1e10-scaled pair).
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import paropt_amd as pa

    c = pa.Context(0)
    yield c
    c.close()


def qn_pair(k, n):
    from oracle import paropt_oracle as po

    idx = np.arange(n, dtype=np.uint64)
    sv = 2.0 * po.u01(0, 1000 + k, idx) - 1.0
    h = 0.5 + 4.0 * po.u01(0, 5, idx)
    noise = 0.2 * (2.0 * po.u01(0, 2000 + k, idx) - 1.0)
    yv = h * sv + noise
    if k % 5 == 4:
        yv = -0.3 * h * sv + noise
    if k % 7 == 6:
        yv = 1e10 * noise
    return sv, yv


@pytest.mark.parametrize("name", golden_names("qn_"))
def test_quasi_newton_golden(ctx, name):
    import paropt_amd as pa
    from oracle import paropt_oracle as po

    g, case = load_golden(name)
    a = case["args"]
    n, msub, steps = int(g["n"][0]), int(g["msub_max"][0]), int(g["steps"][0])
    if a["type"] == "bfgs":
        qn = pa.LBFGS(ctx, n, msub, "damped_update" if a["update"] == "damped" else "skip_negative_curvature")
    else:
        qn = pa.LSR1(ctx, n, msub)
    qn.setInitDiagonalType(a.get("diag", "yty_over_yts"))
    xp_np = -1.0 + 2.0 * po.u01(0, 7, np.arange(n, dtype=np.uint64))
    xp = pa.PVec(ctx, n).from_numpy(xp_np)
    s, y, out = pa.PVec(ctx, n), pa.PVec(ctx, n), pa.PVec(ctx, n)
    mtol = 1e-8 if a["type"] == "bfgs" else 5e-6
    for k in range(steps):
        sn, yn = qn_pair(k, n)
        s.from_numpy(sn)
        y.from_numpy(yn)
        rc = qn.update(s, y)
        p = "k%02d/" % k
        assert rc == int(g[p + "rc"][0]), "update return code at step %d" % k
        b0, d0, M, Z = qn.getCompactMat()
        assert len(Z) == int(g[p + "size"][0])
        assert abs(b0 - g[p + "b0"][0]) <= 1e-11 * abs(b0)
        if len(Z):
            np.testing.assert_allclose(d0, g[p + "d0"], rtol=1e-11)
            Mref = g[p + "M"].reshape(len(Z), len(Z)).T
            np.testing.assert_allclose(M, Mref, rtol=1e-10, atol=1e-11 * np.abs(Mref).max())
        qn.mult(xp, out)
        scale = np.abs(g[p + "mult"]).max()
        np.testing.assert_allclose(out.to_numpy(), g[p + "mult"], rtol=0, atol=mtol * scale)
        out.copyValues(s)
        qn.multAdd(-0.5, xp, out)
        fp2 = np.array([out.norm(), out.dot(xp)])
        np.testing.assert_allclose(fp2, g[p + "multadd_fp"], rtol=mtol)
    # s and y are inputs only
    np.testing.assert_array_equal(s.to_numpy(), sn)
    np.testing.assert_array_equal(y.to_numpy(), yn)


def test_compact_identity(ctx):
    """Reference-independent KAT (examples/limited_memory_test/limited_memory_test.py:80-148):
    with subspace >= number of pairs and y = A s, the compact L-BFGS matrix applied to the last s
    reproduces y (secant equation)."""
    import paropt_amd as pa

    n, m = 40, 12
    rng = np.random.default_rng(0)
    Q, _ = np.linalg.qr(rng.standard_normal((n, n)))
    A = Q @ np.diag(np.linspace(1.0, 10.0, n)) @ Q.T
    for cls in (pa.LBFGS, pa.LSR1):
        qn = cls(ctx, n, m)
        s, y, out = pa.PVec(ctx, n), pa.PVec(ctx, n), pa.PVec(ctx, n)
        for _ in range(m):
            sn = rng.standard_normal(n)
            s.from_numpy(sn)
            y.from_numpy(A @ sn)
            assert qn.update(s, y) == 0
            qn.mult(s, out)
            np.testing.assert_allclose(out.to_numpy(), A @ sn, rtol=0, atol=1e-8 * np.abs(A @ sn).max())
