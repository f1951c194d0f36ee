"""
GPU parity of the vector kernels (through the C ABI) against the golden vectors of the compiled
reference and against the numpy oracle on the same hash-seeded inputs.

Tolerance: fp64 reductions re-associate (two-stage tree vs BLAS order), bound
|err| <= 4 * eps * sqrt(n) * sum|terms|, written below as 1e-13 * n relative to O(1) data;
element-wise results (axpy/scale/copy/fill) are bit-exact.
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


def hvec(ctx, n, aid, seed=0, scale=2.0, shift=-1.0, offset=0):
    import paropt_amd as pa

    return pa.PVec(ctx, n).fill_hash(seed, aid, offset, scale, shift)


def hnp(n, aid, seed=0, scale=2.0, shift=-1.0, offset=0):
    from oracle import paropt_oracle as po

    return shift + scale * po.u01(seed, aid, np.arange(offset, offset + n, dtype=np.uint64))


def test_hash_fill_bit_exact(ctx):
    for n in (1, 2, 63, 1000, 4097):
        v = hvec(ctx, n, 10, seed=3, offset=12345)
        np.testing.assert_array_equal(v.to_numpy(), hnp(n, 10, seed=3, offset=12345))


@pytest.mark.parametrize("name", [n for n in golden_names("vecops_") if "_r" not in n])
def test_vecops_golden(ctx, name):
    g, case = load_golden(name)
    n, nv = int(g["n"][0]), int(g["nvecs"][0])
    x, y = hvec(ctx, n, 10), hvec(ctx, n, 11)
    V = [hvec(ctx, n, 20 + j) for j in range(nv)]
    tol = 1e-13 * n
    assert abs(x.dot(y) - g["dot"][0]) <= tol
    assert abs(x.norm() - g["norm"][0]) <= tol
    assert x.maxabs() == g["maxabs"][0]
    assert abs(x.l1norm() - g["l1norm"][0]) <= tol
    np.testing.assert_allclose(x.mdot(V), g["mdot"], rtol=0, atol=tol)
    y.scale(0.75)
    y.axpy(-1.25, x)
    assert abs(y.norm() - g["post_norm"][0]) <= tol
    assert abs(y.l1norm() - g["post_l1"][0]) <= tol
    assert abs(y.dot(x) - g["post_dot"][0]) <= tol
    if "post_y" in g:
        np.testing.assert_allclose(y.to_numpy(), g["post_y"], rtol=1e-15, atol=1e-15)


@pytest.mark.parametrize("n", [1, 2, 3, 255, 256, 257, 511, 513, 100003, (1 << 20) + 3])
@pytest.mark.parametrize("nv", [1, 4, 5, 8, 9, 16, 17, 32, 33, 40, 72])
def test_mdot_vs_oracle(ctx, n, nv):
    if n > 200000 and nv not in (1, 33, 72):
        pytest.skip("large n covered for a subset of widths")
    x = hvec(ctx, n, 10)
    V = [hvec(ctx, n, 20 + j) for j in range(nv)]
    xn = hnp(n, 10)
    ref = np.array([np.dot(xn, hnp(n, 20 + j)) for j in range(nv)])
    np.testing.assert_allclose(x.mdot(V), ref, rtol=0, atol=1e-13 * max(n, 16))


@pytest.mark.parametrize("nv", [96, 97, 200])
def test_mdot_wider_than_one_panel(ctx, nv):
    """The reference's mdot has no limit on the number of vectors (src/ParOptVec.cpp:152-170); one kernel launch
    takes 96 columns, wider calls are processed in slabs behind the same entry point."""
    n = 1237
    x = hvec(ctx, n, 10)
    V = [hvec(ctx, n, 20 + j) for j in range(nv)]
    xn = hnp(n, 10)
    ref = np.array([np.dot(xn, hnp(n, 20 + j)) for j in range(nv)])
    np.testing.assert_allclose(x.mdot(V), ref, rtol=0, atol=1e-13 * n)


def test_elementwise_bit_exact(ctx):
    import paropt_amd as pa

    for n in (1, 7, 1000, 4097):
        x, y = hvec(ctx, n, 10), hvec(ctx, n, 11)
        xn, yn = hnp(n, 10), hnp(n, 11)
        z = pa.PVec(ctx, n)
        np.testing.assert_array_equal(z.to_numpy(), np.zeros(n))  # zero-initialised like ParOptBasicVec
        z.copyValues(x)
        np.testing.assert_array_equal(z.to_numpy(), xn)
        z.set(0.5)
        np.testing.assert_array_equal(z.to_numpy(), np.full(n, 0.5))
        z.zeroEntries()
        np.testing.assert_array_equal(z.to_numpy(), np.zeros(n))
        y.axpy(-1.25, x)
        # fma contraction on the device: one rounding instead of two
        np.testing.assert_allclose(y.to_numpy(), yn + (-1.25) * xn, rtol=0, atol=2e-16 * 3)
        V = [hvec(ctx, n, 20 + j) for j in range(5)]
        al = np.array([0.3, -1.5, 2.0, 0.0, 1e-3])
        z.copyValues(x)
        z.maxpy(0.5, al, V)
        ref = 0.5 * xn + sum(al[j] * hnp(n, 20 + j) for j in range(5))
        np.testing.assert_allclose(z.to_numpy(), ref, rtol=0, atol=1e-15 * 8)


def test_get_array_roundtrip(ctx):
    import paropt_amd as pa

    v = pa.PVec(ctx, 1001)
    a = v.getArray()
    a[:] = np.arange(1001.0)
    v.syncToDevice()
    assert v.l1norm() == 1000 * 1001 / 2
    v.scale(2.0)
    np.testing.assert_array_equal(v.to_numpy(), 2.0 * np.arange(1001.0))


def test_empty_vector(ctx):
    """Rank-local size 0 is legal in the reference (getArray may return NULL, src/ParOptVec.cpp:212)."""
    import paropt_amd as pa

    v, w = pa.PVec(ctx, 0), pa.PVec(ctx, 0)
    assert v.dot(w) == 0.0 and v.norm() == 0.0 and v.maxabs() == 0.0 and v.l1norm() == 0.0
    np.testing.assert_array_equal(v.mdot([w, w]), np.zeros(2))


def test_size_mismatch_is_an_error(ctx):
    import paropt_amd as pa

    with pytest.raises(pa.ParOptAMDError):
        pa.PVec(ctx, 10).dot(pa.PVec(ctx, 11))


@pytest.mark.parametrize("n", [1, 5, 127, 128, 129, 1000, 70001])
@pytest.mark.parametrize("nv", [1, 5, 16, 17, 32, 42, 48, 49, 72, 80])
def test_wgram_vs_numpy(ctx, n, nv):
    """W = P^T diag(d) P on the fp64 MFMA path; asymmetric data catches row/col swaps."""
    import paropt_amd as pa

    if n > 10000 and nv not in (5, 42, 80):
        pytest.skip("large n covered for a subset of widths")
    d = hvec(ctx, n, 9, scale=1.0, shift=0.5)
    V = [hvec(ctx, n, 20 + j, scale=2.0, shift=-1.0 + 0.1 * j) for j in range(nv)]
    dn = hnp(n, 9, scale=1.0, shift=0.5)
    P = np.stack([hnp(n, 20 + j, scale=2.0, shift=-1.0 + 0.1 * j) for j in range(nv)], axis=1)
    ref = P.T @ (dn[:, None] * P)
    W = pa.wgram(d, V)
    np.testing.assert_allclose(W, ref, rtol=0, atol=1e-13 * max(n, 64) * 10)
    np.testing.assert_array_equal(W, W.T)


@pytest.mark.parametrize("n", [1, 129, 1000, 70001])
@pytest.mark.parametrize("nv", [81, 96, 97, 128, 200])
def test_wgram_wider_than_one_launch(ctx, n, nv):
    """Panels beyond one launch's 80 columns go by column-block pairs (wgram.hip: k_wgram): the reference has no
    limit on ncon or on the quasi-Newton width (src/ParOptInteriorPoint.cpp:1935-1950, 2648-2654)."""
    import paropt_amd as pa

    d = hvec(ctx, n, 9, scale=1.0, shift=0.5)
    V = [hvec(ctx, n, 20 + j, scale=2.0, shift=-1.0 + 0.01 * j) for j in range(nv)]
    dn = hnp(n, 9, scale=1.0, shift=0.5)
    P = np.stack([hnp(n, 20 + j, scale=2.0, shift=-1.0 + 0.01 * j) for j in range(nv)], axis=1)
    ref = P.T @ (dn[:, None] * P)
    W = pa.wgram(d, V)
    np.testing.assert_allclose(W, ref, rtol=0, atol=1e-13 * max(n, 64) * 10)
    np.testing.assert_array_equal(W, W.T)


@pytest.mark.parametrize("nv", [97, 150, 200])
def test_panel_launchers_wider_than_their_tables(ctx, nv):
    """mdot / maxpy beyond the 96-entry kernel argument tables are slabbed inside the launchers."""
    import paropt_amd as pa

    n = 5003
    x = hvec(ctx, n, 3)
    V = [hvec(ctx, n, 40 + j, scale=2.0, shift=-1.0) for j in range(nv)]
    xn = hnp(n, 3)
    P = np.stack([hnp(n, 40 + j, scale=2.0, shift=-1.0) for j in range(nv)], axis=1)
    np.testing.assert_allclose(x.mdot(V), P.T @ xn, rtol=0, atol=1e-13 * n * 10)
    alpha = np.linspace(-1.0, 1.0, nv)
    y = hvec(ctx, n, 4)
    y.maxpy(0.5, alpha, V)
    np.testing.assert_allclose(y.to_numpy(), 0.5 * hnp(n, 4) + P @ alpha, rtol=0, atol=1e-12)


@pytest.mark.parametrize("n", [1, 129, 1000, 70001])
@pytest.mark.parametrize("nv", [2, 9, 16, 17, 33, 43, 48, 49, 80])
def test_wgram_with_preweighted_rhs_column(ctx, n, nv):
    """The last column t is pre-weighted: its row/column hold the plain dots P^T t, the rest is the weighted Gram."""
    import paropt_amd as pa

    d = hvec(ctx, n, 9, scale=1.0, shift=0.5)
    V = [hvec(ctx, n, 20 + j, scale=2.0, shift=-1.0 + 0.1 * j) for j in range(nv)]
    dn = hnp(n, 9, scale=1.0, shift=0.5)
    P = np.stack([hnp(n, 20 + j, scale=2.0, shift=-1.0 + 0.1 * j) for j in range(nv)], axis=1)
    ref = P.T @ (dn[:, None] * P)
    ref[:, nv - 1] = P.T @ P[:, nv - 1]
    ref[nv - 1, :] = ref[:, nv - 1]
    W = pa.wgram(d, V, rhs_last=True)
    np.testing.assert_allclose(W, ref, rtol=0, atol=1e-13 * max(n, 64) * 10)
    np.testing.assert_array_equal(W, W.T)


@pytest.mark.parametrize("n", [512, 513, 575, 641, 4097, 70001, 300007])
@pytest.mark.parametrize("nv,rhs_last", [(65, False), (68, True), (69, False), (72, False), (73, True), (76, False),
                                         (77, True), (80, False)])
def test_wgram_wide_panel_producer_consumer(ctx, n, nv, rhs_last):
    """Panels of 65-80 columns (config 3's data with an L-BFGS(20) memory: 32 + 40 + the pre-weighted column = 73) take
    the 64-row producer/consumer kernel (wgram.hip: wgram_pc64_kernel) from n = 512 on: tile edges (n mod 64 = 0, 1,
    63), ragged last column groups, the pre-weighted column in the last group, and more tiles than workgroups."""
    import paropt_amd as pa

    if n > 100000 and nv not in (73, 80):
        pytest.skip("largest n for the bench's width and the widest panel")
    d = hvec(ctx, n, 9, scale=1.0, shift=0.5)
    V = [hvec(ctx, n, 20 + j, scale=2.0, shift=-1.0 + 0.02 * j) for j in range(nv)]
    dn = hnp(n, 9, scale=1.0, shift=0.5)
    P = np.stack([hnp(n, 20 + j, scale=2.0, shift=-1.0 + 0.02 * j) for j in range(nv)], axis=1)
    ref = P.T @ (dn[:, None] * P)
    if rhs_last:
        ref[:, nv - 1] = P.T @ P[:, nv - 1]
        ref[nv - 1, :] = ref[:, nv - 1]
    W = pa.wgram(d, V, rhs_last=rhs_last)
    np.testing.assert_allclose(W, ref, rtol=0, atol=1e-13 * max(n, 64) * 10)
    np.testing.assert_array_equal(W, W.T)


def _group_sums(dn, P, nwcon, nw, skip, alpha):
    """alpha * (rounded products d * p added in index order over each group): the arithmetic of both kernels."""
    period = nw + skip
    U = np.zeros((nwcon, P.shape[1]))
    base = np.arange(nwcon) * period
    for k in range(nw):
        U = U + dn[base + k, None] * P[base + k, :]
    return alpha * U


@pytest.mark.parametrize("n,nwcon,nw,skip", [
    (2000, 100, 20, 0),      # config 4's pattern: the groups cover every variable
    (2001, 100, 20, 0),      # odd length
    (3000, 100, 20, 0),      # ordinary tiles behind the group tiles
    (70001, 3332, 20, 1),    # odd period: an even number of groups per tile
    (70001, 9000, 7, 0),
    (5000, 190, 24, 2),      # the widest group the fused form takes
    (5000, 160, 11, 20),
    (4001, 2000, 2, 0),      # 64 groups per tile, the last variable outside every group
    (600, 25, 24, 0),        # a handful of tiles only
])
@pytest.mark.parametrize("nv,rhs_last", [(4, False), (5, True), (13, True), (24, False), (25, True), (36, True)])
def test_wgram_with_structured_panel_image(ctx, n, nwcon, nw, skip, nv, rhs_last):
    """The structured sparse-Jacobian panel image U = alpha Aw (d o P) riding in the Gram pass (wgram.hip: GramGeom):
    U has the bits of the stand-alone kernel (and of the same sums in numpy), W is the weighted Gram."""
    import paropt_amd as pa

    d = hvec(ctx, n, 9, scale=1.0, shift=0.5)
    V = [hvec(ctx, n, 20 + j, scale=2.0, shift=-1.0 + 0.1 * j) for j in range(nv)]
    dn = hnp(n, 9, scale=1.0, shift=0.5)
    P = np.stack([hnp(n, 20 + j, scale=2.0, shift=-1.0 + 0.1 * j) for j in range(nv)], axis=1)
    ncols = nv - 1 if rhs_last else nv
    U = [pa.PVec(ctx, nwcon) for _ in range(ncols)]
    U2 = [pa.PVec(ctx, nwcon) for _ in range(ncols)]
    for u in U:
        u.set(123.0)
    W, fused = pa.wgram_with_groups(d, V, nwcon, nw, skip, -1.0, U, rhs_last=rhs_last)
    ref = P.T @ (dn[:, None] * P)
    if rhs_last:
        ref[:, nv - 1] = P.T @ P[:, nv - 1]
        ref[nv - 1, :] = ref[:, nv - 1]
    np.testing.assert_allclose(W, ref, rtol=0, atol=1e-13 * max(n, 64) * 10)
    np.testing.assert_array_equal(W, W.T)
    # panels of fewer than 21 columns keep the single-role Gram and the stand-alone panel image (faster there)
    assert fused == ((nv + 3) // 4 >= 6), "the fused kernel covers this shape from six column groups on"
    pa.group_panel(d, V[:ncols], nwcon, nw, skip, -1.0, U2)
    want = _group_sums(dn, P[:, :ncols], nwcon, nw, skip, -1.0)
    for j in range(ncols):
        np.testing.assert_array_equal(U2[j].to_numpy(), want[:, j])
        if fused:
            np.testing.assert_array_equal(U[j].to_numpy(), want[:, j])
        else:
            assert (U[j].to_numpy() == 123.0).all()


@pytest.mark.parametrize("n,nwcon,nw,skip,nv", [(300, 10, 20, 0, 5),      # below the producer/consumer form's size
                                                (5000, 30, 129, 0, 5),    # a group wider than a tile
                                                (5000, 100, 25, 0, 5),    # ... than the sums' register budget
                                                (5000, 100, 21, 0, 5),    # fewer than six column groups: single-role Gram + stand-alone image
                                                (5000, 100, 20, 0, 44)])  # wider than the instantiations that carry it
def test_wgram_with_groups_declines_what_it_does_not_cover(ctx, n, nwcon, nw, skip, nv):
    import paropt_amd as pa

    d = hvec(ctx, n, 9, scale=1.0, shift=0.5)
    V = [hvec(ctx, n, 20 + j, scale=2.0, shift=-1.0 + 0.1 * j) for j in range(nv)]
    U = [pa.PVec(ctx, nwcon) for _ in range(nv)]
    for u in U:
        u.set(123.0)
    W, fused = pa.wgram_with_groups(d, V, nwcon, nw, skip, -1.0, U)
    assert not fused
    np.testing.assert_array_equal(W, pa.wgram(d, V))
    assert all((u.to_numpy() == 123.0).all() for u in U)


def test_live_mdot_timing_hook(ctx):
    """po_ctx_time_mdot: launches of exactly the requested width are timed with HIP events, others are not."""
    import paropt_amd as pa

    n = 100_000
    x = pa.PVec(ctx, n).fill_hash(0, 1, 0, 1.0, 0.0)
    V = [pa.PVec(ctx, n).fill_hash(0, 10 + j, 0, 1.0, 0.0) for j in range(5)]
    ctx.time_mdot(5)
    ref = x.mdot(V)
    x.mdot(V[:3])
    x.mdot(V)
    ms, cnt = ctx.time_mdot_result()
    assert cnt == 2 and ms > 0.0
    ctx.time_mdot(0)
    np.testing.assert_array_equal(x.mdot(V), ref)
    assert ctx.time_mdot_result() == (0.0, 0)
    red, launches = ctx.counters()
    assert red > 0 and launches > 0
