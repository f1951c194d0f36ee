"""Shared helpers for the CSR sparse-constraint tests: pattern generators and a numpy emulation of the
device factorization (same storage, same level order, same arithmetic per entry) used to validate the
host analysis without a GPU."""
import numpy as np


def chain_pattern(n, span=2, stride=1, reverse=False):
    rows = (n - span) // stride + 1 if n >= span else 0
    rowp = np.arange(rows + 1, dtype=np.intc) * span
    cols = np.zeros(rows * span, dtype=np.intc)
    for k in range(span):
        cols[(span - 1 - k if reverse else k)::span] = np.arange(rows) * stride + k
    return rowp, cols


def random_pattern(n, w, max_row, seed, local=None):
    rng = np.random.default_rng(seed)
    rowp = [0]
    cols = []
    for i in range(w):
        ln = int(rng.integers(0, max_row + 1))
        if local:
            base = int(rng.integers(0, n))
            cand = np.unique((base + rng.integers(0, local, size=ln)) % n)
        else:
            cand = np.unique(rng.integers(0, n, size=ln))
        cand = rng.permutation(cand)
        cols.extend(int(c) for c in cand)
        rowp.append(len(cols))
    return np.array(rowp, dtype=np.intc), np.array(cols, dtype=np.intc)


def grid_pattern(nx, ny):
    """One constraint per grid edge of an nx x ny grid of variables (rows of 2): S is a line-graph Laplacian."""
    idx = lambda i, j: i * ny + j
    rowp, cols = [0], []
    for i in range(nx):
        for j in range(ny):
            if i + 1 < nx:
                cols += [idx(i, j), idx(i + 1, j)]
                rowp.append(len(cols))
            if j + 1 < ny:
                cols += [idx(i, j + 1), idx(i, j)]
                rowp.append(len(cols))
    return np.array(rowp, dtype=np.intc), np.array(cols, dtype=np.intc)


def dense_jacobian(n, rowp, cols, data):
    w = len(rowp) - 1
    A = np.zeros((w, n))
    for i in range(w):
        for p in range(rowp[i], rowp[i + 1]):
            A[i, cols[p]] += data[p]
    return A


def emulate_factor(sym, S):
    """Row Cholesky on the symbolic pattern in the device's schedule; returns L (dense, permuted space).
    Ordinary rows of a level may only read earlier levels; the rows of a front may also read the earlier rows of
    the same front, and only through the dense tail of the row."""
    w = len(sym.perm)
    P = sym.perm
    Sp = S[np.ix_(P, P)]
    Lp, Lc = sym.Lrowp, sym.Lcols
    L = np.zeros((w, w))
    done = np.zeros(w, dtype=bool)
    for lev in range(sym.nlevels):
        rows = np.arange(sym.level_ptr[lev], sym.level_ptr[lev + 1])
        for i in rows:
            f0 = sym.front_of[i]
            for p in range(Lp[i], Lp[i + 1]):
                j = Lc[p]
                if j < i:
                    in_front = f0 >= 0 and j >= f0
                    assert done[j] or in_front, "row %d needs row %d which is not in an earlier level" % (i, j)
                    if in_front:  # dense tail: columns f0 .. i, contiguous at the end of the row
                        assert p == Lp[i + 1] - 1 - (i - j)
                    L[i, j] = (Sp[i, j] - L[i, :j] @ L[j, :j]) / L[j, j]
                else:
                    assert j == i and p == Lp[i + 1] - 1
                    L[i, i] = np.sqrt(Sp[i, i] - L[i, :i] @ L[i, :i])
        done[rows] = True
    assert done.all()
    return L, Sp
