"""CPU: the one-time host analysis of the CSR sparse-constraint path (paropt_amd/csrc/csr.cpp) through the
C ABI (po_csr_symbolic_*): ordering is a permutation, the factor pattern holds all fill, the level sets
respect every dependency, and a numpy emulation of the device factorization on that structure reproduces
S = L L^T.  No device needed."""
import numpy as np
import pytest

from paropt_amd import CsrSymbolic
from paropt_amd.lib import ParOptAMDError

from csr_helpers import chain_pattern, dense_jacobian, emulate_factor, grid_pattern, random_pattern

CASES = {
    "chain2": lambda: (300, *chain_pattern(300, 2, 1)),
    "chain5_rev": lambda: (257, *chain_pattern(257, 5, 2, reverse=True)),
    "block_diag": lambda: (240, *chain_pattern(240, 4, 4)),
    "grid": lambda: (12 * 11, *grid_pattern(12, 11)),
    "grid_fronts": lambda: (26 * 24, *grid_pattern(26, 24)),
    "random_local": lambda: (400, *random_pattern(400, 250, 5, 1, local=12)),
    "random_global": lambda: (150, *random_pattern(150, 90, 3, 2)),
    "empty_rows": lambda: (50, np.array([0, 0, 2, 2, 3, 3], dtype=np.intc), np.array([4, 1, 4], dtype=np.intc)),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_symbolic_structure(name):
    n, rowp, cols = CASES[name]()
    w = len(rowp) - 1
    sym = CsrSymbolic(n, rowp, cols)
    assert sorted(sym.perm.tolist()) == list(range(w))
    assert sym.nnz == rowp[-1]
    rng = np.random.default_rng(0)
    data = rng.uniform(0.5, 1.5, size=max(int(rowp[-1]), 1))[: rowp[-1]]
    A = dense_jacobian(n, rowp, cols, data)
    d = rng.uniform(0.5, 2.0, size=n)
    c = rng.uniform(0.1, 1.0, size=w)
    S = np.diag(c) + (A * d) @ A.T
    assert sym.nnzS == np.count_nonzero(np.tril((np.abs(A) @ np.abs(A).T) + np.eye(w)))
    # rows sorted, diagonal last, parents above children
    for i in range(w):
        r = sym.Lcols[sym.Lrowp[i]:sym.Lrowp[i + 1]]
        assert r[-1] == i and np.all(np.diff(r) > 0)
        assert sym.parent[i] == -1 or sym.parent[i] > i
    L, Sp = emulate_factor(sym, S)
    np.testing.assert_allclose(L @ L.T, Sp, rtol=1e-12, atol=1e-12)
    # the exact factor has no entry outside the symbolic pattern
    Lref = np.linalg.cholesky(Sp)
    mask = np.zeros((w, w), dtype=bool)
    for i in range(w):
        mask[i, sym.Lcols[sym.Lrowp[i]:sym.Lrowp[i + 1]]] = True
    assert np.all(np.abs(Lref[~mask]) < 1e-13)
    np.testing.assert_allclose(L, Lref, rtol=1e-10, atol=1e-12)
    # the same level sets, descending, schedule the backward solve: every row below the diagonal in column j
    # sits in a LATER level than j
    level = np.zeros(w, dtype=int)
    for lev in range(sym.nlevels):
        level[sym.level_ptr[lev]:sym.level_ptr[lev + 1]] = lev
    ii, jj = np.nonzero(np.tril(mask, -1))
    same_front = (sym.front_of[ii] >= 0) & (sym.front_of[ii] == sym.front_of[jj])
    assert np.all((level[ii] > level[jj]) | same_front)
    if name == "grid_fronts":
        assert sym.nfronts > 0 and sym.max_front >= 16


def test_nested_dissection_keeps_chains_shallow():
    # natural order on a chain needs w levels; the dissection ordering must stay near log2(w) + leaf size
    n = 20000
    rowp, cols = chain_pattern(n, 2, 1)
    sym = CsrSymbolic(n, rowp, cols)
    assert sym.nlevels < 40, sym.nlevels
    assert sym.nnzL < 3 * sym.nnzS


def test_fronts_flatten_grid_like_patterns(monkeypatch):
    # a 2-D pattern: without fronts every separator vertex is a level of its own
    rowp, cols = grid_pattern(60, 60)
    sym = CsrSymbolic(3600, rowp, cols)
    monkeypatch.setenv("PAROPT_AMD_NO_FRONTS", "1")
    ref = CsrSymbolic(3600, rowp, cols)
    assert ref.nfronts == 0 and sym.nfronts > 10
    assert sym.nnzL == ref.nnzL and sym.nlevels * 4 < ref.nlevels, (sym.nlevels, ref.nlevels)


def test_block_diagonal_has_no_fill():
    rowp, cols = chain_pattern(4000, 4, 4)
    sym = CsrSymbolic(4000, rowp, cols)
    assert sym.nnzL == sym.nnzS == 1000 and sym.nlevels == 1 and sym.sorted_input


def test_bad_patterns_are_rejected():
    with pytest.raises(ParOptAMDError):
        CsrSymbolic(5, np.array([0, 2], dtype=np.intc), np.array([1, 1], dtype=np.intc))  # duplicate
    with pytest.raises(ParOptAMDError):
        CsrSymbolic(5, np.array([0, 1], dtype=np.intc), np.array([5], dtype=np.intc))  # out of range
    with pytest.raises(ParOptAMDError):
        CsrSymbolic(5, np.array([0, 2, 1], dtype=np.intc), np.array([0, 1], dtype=np.intc))  # rowp decreasing


def test_symbolic_random_soak():
    """Twenty random patterns (grids with extra random rows, local and global random rows): the schedule's
    dependency rules hold and the emulated factorization reproduces S."""
    rng = np.random.default_rng(7)
    for trial in range(20):
        if trial % 2 == 0:
            nx, ny = int(rng.integers(4, 26)), int(rng.integers(4, 26))
            n = nx * ny
            rowp, cols = grid_pattern(nx, ny)
            ep, ec = random_pattern(n, int(rng.integers(0, 20)), 6, int(rng.integers(1 << 30)))
            rowp = np.concatenate([rowp, rowp[-1] + ep[1:]]).astype(np.intc)
            cols = np.concatenate([cols, ec]).astype(np.intc)
        else:
            n = int(rng.integers(20, 300))
            rowp, cols = random_pattern(n, int(rng.integers(1, 250)), int(rng.integers(1, 12)),
                                        int(rng.integers(1 << 30)), local=int(rng.integers(4, 50)))
        w = len(rowp) - 1
        sym = CsrSymbolic(n, rowp, cols)
        data = rng.uniform(0.5, 1.5, size=int(rowp[-1]))
        A = dense_jacobian(n, rowp, cols, data)
        S = np.diag(rng.uniform(0.1, 1.0, size=w)) + (A * rng.uniform(0.5, 2.0, size=n)) @ A.T
        L, Sp = emulate_factor(sym, S)
        np.testing.assert_allclose(L @ L.T, Sp, rtol=1e-11, atol=1e-11)
