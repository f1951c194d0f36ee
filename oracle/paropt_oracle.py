"""
TEST INFRASTRUCTURE ONLY -- CPU restatement ("oracle") of the ParOpt interior-point hot path.

This module restates, in plain numpy, the algorithm of the reference implementation
(smdogroup/paropt v2.1.5) for the path named by BASELINE.json:

  * the distributed vector reductions      src/ParOptVec.cpp:63-204
  * compact L-BFGS / L-SR1                 src/ParOptQuasiNewton.cpp:162-459, 636-809
  * the interior-point iteration (w = 0)   src/ParOptInteriorPoint.cpp:1337-5656

It follows the reference's *operation order* ("mirror mode"), so that it is an
independent check of the fused HIP product in paropt_amd/.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import it; the product
never does.

Parity pin: every function here is checked against golden vectors produced by the real
reference compiled from /root/reference (oracle/Makefile -> oracle/_ref/ref_driver,
oracle/make_golden.py -> tests/golden/*.npz); see tests/test_oracle_golden.py.

The small dense factorizations use scipy.linalg.lu_factor/lu_solve, i.e. LAPACK
dgetrf/dgetrs -- the routines the reference calls (src/ParOptBlasLapack.h:29-30).
"""
import math

import numpy as np
import scipy.linalg as sla

# --------------------------------------------------------------------------------------
# Counter-hash synthetic data (DESIGN.md "Synthetic data"): u01(seed, array id, global i)
# --------------------------------------------------------------------------------------
_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(z):
    z = np.asarray(z, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = z + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def u01(seed, aid, idx):
    """U[0,1) as a pure function of (seed, array id, global index)."""
    idx = np.asarray(idx, dtype=np.uint64)
    with np.errstate(over="ignore"):
        base = np.uint64(seed) * np.uint64(0x9E3779B97F4A7C15) + np.uint64(aid) * np.uint64(
            0xD1B54A32D192ED03
        )
        h = splitmix64(base + idx)
    return (h >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def shard(n, rank, size):
    """Contiguous row blocks: (nlocal, offset); the first n % size ranks get one extra."""
    base, rem = divmod(int(n), int(size))
    nlocal = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return nlocal, offset


# --------------------------------------------------------------------------------------
# Communicators: the reference's MPI_Allreduce sites
# --------------------------------------------------------------------------------------
class SelfComm:
    rank = 0
    size = 1

    def allreduce(self, arr, op="sum"):
        return np.array(arr, dtype=np.float64, copy=True)


class TorchComm:
    """torch.distributed (gloo on CPU) stand-in for the MPI communicator."""

    def __init__(self):
        import torch.distributed as dist

        self.dist = dist
        self.rank = dist.get_rank()
        self.size = dist.get_world_size()

    def allreduce(self, arr, op="sum"):
        import torch

        t = torch.tensor(np.atleast_1d(np.asarray(arr, dtype=np.float64)))
        ops = {
            "sum": self.dist.ReduceOp.SUM,
            "min": self.dist.ReduceOp.MIN,
            "max": self.dist.ReduceOp.MAX,
        }
        self.dist.all_reduce(t, op=ops[op])
        return t.numpy().copy()


# --------------------------------------------------------------------------------------
# Vector reductions -- src/ParOptVec.cpp
# --------------------------------------------------------------------------------------
class VecOps:
    def __init__(self, comm=None):
        self.comm = comm if comm is not None else SelfComm()

    def dot(self, x, y):  # src/ParOptVec.cpp:124-143
        return float(self.comm.allreduce([np.dot(x, y)])[0])

    def mdot(self, x, vecs):  # src/ParOptVec.cpp:152-170
        loc = np.array([np.dot(x, v) for v in vecs], dtype=np.float64)
        if len(vecs) == 0:
            return loc
        return self.comm.allreduce(loc)

    def norm(self, x):  # src/ParOptVec.cpp:63-80 (dnrm2, squared, summed, sqrt)
        return math.sqrt(float(self.comm.allreduce([np.dot(x, x)])[0]))

    def maxabs(self, x):  # src/ParOptVec.cpp:87-99
        loc = float(np.max(np.abs(x))) if x.size else 0.0
        return float(self.comm.allreduce([loc], "max")[0])

    def l1norm(self, x):  # src/ParOptVec.cpp:106-116
        return float(self.comm.allreduce([np.sum(np.abs(x))])[0])


# --------------------------------------------------------------------------------------
# Compact quasi-Newton -- src/ParOptQuasiNewton.cpp
# --------------------------------------------------------------------------------------
class _CompactQN:
    """Shared storage/rotation logic of ParOptLBFGS and ParOptLSR1."""

    def __init__(self, n, msub_max, ops):
        self.ops = ops
        self.n = n
        self.msub_max = msub_max
        self.S = [np.zeros(n) for _ in range(msub_max)]
        self.Y = [np.zeros(n) for _ in range(msub_max)]
        self.diag_type = "yty_over_yts"
        self.reset()

    def update_mult(self, x, z, zw):  # update(x, z, zw): no-op in the base classes (.h:60-63)
        return 0

    def reset(self):  # src/ParOptQuasiNewton.cpp:127-142 / 603-618
        self.msub = 0
        self.b0 = 1.0
        m = self.msub_max
        self.D = np.zeros(m)
        self.L = np.zeros((m, m))  # L[i, j] = S[i].Y[j], j < i
        self.B = np.zeros((m, m))  # B = S^T S
        self.M = np.zeros((0, 0))
        self.d0 = np.zeros(0)
        self.Z = []
        self.lu = None

    def _store(self, s, y):
        """Append (or rotate in) a pair and refresh B, D, L: :266-321 / :650-705."""
        m = self.msub_max
        if self.msub < m:
            self.S[self.msub][:] = s
            self.Y[self.msub][:] = y
            self.msub += 1
        elif self.msub == m and m > 0:
            self.S[0][:] = s
            self.Y[0][:] = y
            self.S = self.S[1:] + self.S[:1]
            self.Y = self.Y[1:] + self.Y[:1]
            k = self.msub
            self.D[: k - 1] = self.D[1:k].copy()
            self.B[: k - 1, : k - 1] = self.B[1:k, 1:k].copy()
            # only the strictly-lower part is shifted in the reference (:298-302)
            Lold = self.L.copy()
            for i in range(k - 1):
                for j in range(i):
                    self.L[i, j] = Lold[i + 1, j + 1]
        k = self.msub
        dot = self.ops.dot
        for i in range(k):
            self.B[k - 1, i] = dot(self.S[k - 1], self.S[i])
            self.B[i, k - 1] = self.B[k - 1, i]
        if k > 0:
            self.D[k - 1] = dot(self.S[k - 1], self.Y[k - 1])
        for i in range(k - 1):
            self.L[k - 1, i] = dot(self.S[k - 1], self.Y[i])

    def _factor(self):
        if self.M.shape[0] > 0:
            self.lu = sla.lu_factor(self.M, check_finite=False)
        else:
            self.lu = None

    def get_compact(self):
        """(b0, d0, M, Z): src/ParOptQuasiNewton.cpp:471-487 / 821-837."""
        return self.b0, self.d0, self.M, self.Z

    def mult(self, x):  # :390-418 / :760-778
        y = self.b0 * x
        if len(self.Z) > 0:
            rz = self.ops.mdot(x, self.Z)
            rz = rz * self.d0
            rz = sla.lu_solve(self.lu, rz, check_finite=False)
            rz = rz * self.d0
            for i in range(len(self.Z)):
                y = y - rz[i] * self.Z[i]
        return y

    def mult_add(self, alpha, x, y):  # :432-459 / :791-809 (y updated in place)
        y += (self.b0 * alpha) * x
        if len(self.Z) > 0:
            rz = self.ops.mdot(x, self.Z)
            rz = rz * self.d0
            rz = sla.lu_solve(self.lu, rz, check_finite=False)
            rz = rz * self.d0
            for i in range(len(self.Z)):
                y -= (alpha * rz[i]) * self.Z[i]
        return y


class LBFGS(_CompactQN):
    """ParOptLBFGS: B = b0 I - Z diag(d0) M^-1 diag(d0) Z^T, Z = [S, Y]."""

    def __init__(self, n, msub_max, ops, update_type="skip_negative_curvature"):
        self.update_type = update_type
        super().__init__(n, msub_max, ops)

    def max_size(self):
        return 2 * self.msub_max

    def update(self, s, y):  # src/ParOptQuasiNewton.cpp:162-334
        dot = self.ops.dot
        yTy = dot(y, y)
        yTs = dot(y, s)
        sTs = dot(s, s)
        if 1e-8 * yTy >= abs(yTs):
            return 2
        r = self.mult(s)
        sTBs = dot(r, s)
        eps = 1e-12
        if yTs >= eps:
            b0_init = yTs / sTs if self.diag_type == "yts_over_sts" else yTy / yTs
        else:
            b0_init = 0.5 * (abs(yTy / yTs) + abs(yTs / sTs))
        rc = 0
        y_update = None
        if yTs >= 0.01 * sTBs:
            y_update = y
            self.b0 = b0_init
        elif self.update_type == "skip_negative_curvature":
            return 2
        else:  # damped update :241-263
            rc = 1
            theta = 0.8 * sTBs / (sTBs - yTs)
            r = (1.0 - theta) * r
            r = r + theta * y
            y_update = r
            yTy = dot(y_update, y_update)
            yTs = dot(s, y_update)
            self.b0 = yTs / sTs if self.diag_type == "yts_over_sts" else yTy / yTs
        self._store(s, y_update)
        self._mat_update()
        return rc

    def _mat_update(self):  # computeMatUpdate :339-377
        k = self.msub
        M = np.zeros((2 * k, 2 * k))
        M[:k, :k] = self.b0 * self.B[:k, :k]
        for i in range(k):
            for j in range(i):
                M[i, j + k] = self.L[i, j]
                M[j + k, i] = self.L[i, j]
            M[k + i, k + i] = -self.D[i]
        self.M = M
        self.d0 = np.concatenate([np.full(k, self.b0), np.ones(k)])
        self.Z = [self.S[i] for i in range(k)] + [self.Y[i] for i in range(k)]
        self._factor()


class LSR1(_CompactQN):
    """ParOptLSR1: B = b0 I - Z M^-1 Z^T, Z_i = Y_i - b0 S_i."""

    def max_size(self):
        return self.msub_max

    def update(self, s, y):  # src/ParOptQuasiNewton.cpp:636-747
        dot = self.ops.dot
        yTy = dot(y, y)
        sTy = dot(s, y)
        self.b0 = yTy / sTy if sTy > 1e-12 * yTy else 1.0
        self._store(s, y)
        k = self.msub
        M = self.b0 * self.B[:k, :k].copy()
        for i in range(k):
            for j in range(i):
                M[i, j] -= self.L[i, j]
                M[j, i] -= self.L[i, j]
            M[i, i] -= self.D[i]
        self.M = M
        self.Z = [self.Y[i] - self.b0 * self.S[i] for i in range(k)]
        self.d0 = np.ones(k)
        self._factor()
        return 0


# --------------------------------------------------------------------------------------
# Problems (DESIGN.md "Workloads")
# --------------------------------------------------------------------------------------
class SepProblem:
    """Separable restrictions of the reference's example problems.

    quadratic : examples/random_quadratic/random_quadratic.py:30-57,86-88 with A = diag(q)
    convex    : examples/random_convex/random_convex.py:31-71,104-111 with Q = I, Affine = 1e-3 I
    rosenbrock: examples/rosenbrock/rosenbrock.cpp:34-107 (w = 0 variant)
    """

    def __init__(self, kind, n, c, seed=0, eig_min=1.0, eig_max=100.0, comm=None, nwcon=0, nw=0,
                 nwstart=0, nwskip=0, nwineq=-1, chain=None, nwblock=1, bounds_mode=0):
        self.comm = comm if comm is not None else SelfComm()
        # test data for initAndCheckDesignAndBounds (oracle/ref_driver.cpp SepProblem::bounds_mode), by global index
        # gi: bit 1: gi % 7 == 3 -> lb = ub = midpoint; bit 2: gi % 11 == 5 -> x = lb; bit 4: gi % 13 == 6 -> x = ub
        self.bounds_mode = int(bounds_mode)
        # chain = (span, stride): the CSR form (ParOptSparseProblem, src/ParOptProblem.cpp:624-816) with the
        # rank-local overlapping constraints cw_i = 1 - sum_{k<span} x[i*stride+k]^2 >= 0 of
        # oracle/ref_driver.cpp SepCsrProblem (examples/rosenbrock/sparse_rosenbrock.cpp:75-118 is (2, 1))
        self.chain = tuple(chain) if chain else None
        # weighting constraints cw_i = 1 - sum_{k<nw} x[nwstart + i*(nw+nwskip) + k] (the pattern of
        # examples/rosenbrock/rosenbrock.cpp:131-184); disjoint supports, nwblock = 1
        self.nwcon, self.nw, self.nwstart, self.nwskip = int(nwcon), int(nw), int(nwstart), int(nwskip)
        self.nwineq = self.nwcon if nwineq < 0 else int(nwineq)
        # nwblock > 1 (oracle/ref_driver.cpp SepProblem::wgt): blocks of nwblock constraints share one group of
        # nw variables with different weights; Aw D^-1 Aw^T then has dense diagonal blocks, handled through
        # the general (dense S) quasi-definite solve of this oracle
        self.nwblock = int(nwblock)
        self.wwgt = None
        if self.nwcon > 0:
            blk = np.arange(self.nwcon) // self.nwblock
            j0 = self.nwstart + blk * (self.nw + self.nwskip)
            self.widx = (j0[:, None] + np.arange(self.nw)[None, :])  # (w, nw) variable indices
            if self.nwblock > 1:
                k = (np.arange(self.nwcon) % self.nwblock)[:, None]
                j = np.arange(self.nw)[None, :]
                self.wwgt = 1.0 + 0.5 * ((k * (j + 1) + j) % 3)
                self.csr_form = True
        self.kind = kind
        self.nglobal = int(n)
        self.nlocal, self.offset = shard(n, self.comm.rank, self.comm.size)
        if self.chain:
            span, stride = self.chain
            rows = (self.nlocal - span) // stride + 1 if self.nlocal >= span else 0
            self.cidx = (np.arange(rows) * stride)[:, None] + np.arange(span)[None, :]
            self.nwcon = self.nwineq = rows
            self._cw = np.zeros(rows)
            self._jac = np.zeros((rows, span))
        self.c = 2 if kind == "rosenbrock" else int(c)
        self.seed = seed
        idx = np.arange(self.offset, self.offset + self.nlocal, dtype=np.uint64)
        self.idx = idx
        if kind == "quadratic":
            self.q = eig_min + (eig_max - eig_min) * u01(seed, 1, idx)
            self.b = u01(seed, 2, idx)
            self.A = [u01(seed, 100 + j, idx) for j in range(self.c)]
            self.beta = u01(seed, 4, np.arange(self.c, dtype=np.uint64))
        elif kind == "convex":
            self.b = u01(seed, 2, idx)
            self.A = [u01(seed, 100 + j, idx) for j in range(self.c)]
            loc = np.array([np.sum(a) for a in self.A])
            self.beta = 0.25 * self.comm.allreduce(loc)
        elif kind != "rosenbrock":
            raise ValueError(kind)

    def vars_and_bounds(self):
        n = self.nlocal
        if self.kind == "quadratic":
            x, lb, ub = -2.0 + u01(self.seed, 3, self.idx), np.full(n, -5.0), np.full(n, 5.0)
        elif self.kind == "convex":
            x, lb, ub = 0.05 + 0.9 * u01(self.seed, 3, self.idx), np.zeros(n), np.ones(n)
        else:
            x, lb, ub = np.full(n, -1.0), np.full(n, -2.0), np.full(n, 1.0)
        if self.bounds_mode:
            gi = self.idx.astype(np.int64)
            if self.bounds_mode & 1:
                m = gi % 7 == 3
                mid = 0.5 * (lb + ub)
                lb = np.where(m, mid, lb)
                ub = np.where(m, mid, ub)
            if self.bounds_mode & 2:
                x = np.where(gi % 11 == 5, lb, x)
            if self.bounds_mode & 4:
                x = np.where(gi % 13 == 6, ub, x)
        return x, lb, ub

    def sparse_jacobian_dense(self):
        """Aw as a dense (w, n) array from the stored entries (tests and the dense S of the oracle)."""
        A = np.zeros((self.nwcon, self.nlocal))
        if self.chain:
            np.add.at(A, (np.arange(self.nwcon)[:, None], self.cidx), self._jac)
        elif self.nwcon:
            A[np.arange(self.nwcon)[:, None], self.widx] = -1.0 if self.wwgt is None else -self.wwgt
        return A

    def eval_obj_con(self, x):
        c = self.c
        loc = np.zeros(c + 1)
        if self.chain:  # evalSparseObjCon stores the sparse constraint values (.cpp:724-727)
            self._cw = 1.0 - np.sum(x[self.cidx] ** 2, axis=1)
        if self.kind == "quadratic":
            loc[0] = np.sum(0.5 * self.q * x * x + self.b * x)
            for j in range(c):
                loc[1 + j] = np.dot(self.A[j], x)
        elif self.kind == "convex":
            loc[0] = np.sum(self.b * self.b / (1e-3 + x))
            for j in range(c):
                loc[1 + j] = -np.dot(self.A[j], x)
        else:
            loc[0] = np.sum((1.0 - x[:-1]) ** 2 + 100.0 * (x[1:] - x[:-1] ** 2) ** 2)
            loc[1] = -np.sum(x * x)
            loc[2] = np.sum(x[::2])
        tot = self.comm.allreduce(loc)
        cons = tot[1:].copy()
        if self.kind == "rosenbrock":
            cons[0] += 0.25
            cons[1] += 10.0
        else:
            cons += self.beta
        return 0, float(tot[0]), cons

    # sparse-constraint callbacks: src/ParOptProblem.h:225-266
    def eval_sparse_con(self, x):
        if self.nwcon == 0:
            return np.zeros(0)
        if self.chain:  # the values of the LAST evaluation, whatever x is (.cpp:750-760)
            return self._cw.copy()
        if self.wwgt is not None:
            return 1.0 - np.sum(self.wwgt * x[self.widx], axis=1)
        return 1.0 - np.sum(x[self.widx], axis=1)

    def add_sparse_jacobian(self, alpha, px, out):  # out += alpha * Aw px
        if self.chain:  # the entries of the LAST gradient evaluation (.cpp:762-788)
            out += alpha * np.sum(self._jac * px[self.cidx], axis=1)
        elif self.nwcon and self.wwgt is not None:
            out -= alpha * np.sum(self.wwgt * px[self.widx], axis=1)
        elif self.nwcon:
            out -= alpha * np.sum(px[self.widx], axis=1)
        return out

    def add_sparse_jacobian_transpose(self, alpha, pzw, out):  # out += alpha * Aw^T pzw
        if self.chain:
            np.add.at(out, self.cidx, alpha * self._jac * pzw[:, None])
        elif self.nwcon and self.wwgt is not None:
            np.add.at(out, self.widx, -alpha * self.wwgt * pzw[:, None])
        elif self.nwcon:
            out[self.widx] -= alpha * pzw[:, None]
        return out

    def add_sparse_inner_product(self, alpha, cvec, A):  # A += alpha * diag(Aw diag(cvec) Aw^T)
        if self.chain:
            A += alpha * np.sum(self._jac ** 2 * cvec[self.cidx], axis=1)
        elif self.nwcon:
            A += alpha * np.sum(cvec[self.widx], axis=1)
        return A

    def _chain_hessian_diag(self, zw):
        """2 zw_i on the diagonal entries of every chain constraint's variables (Lagrangian f - z.c - zw.cw)."""
        h = np.zeros(self.nlocal)
        if self.chain and zw is not None and len(zw):
            np.add.at(h, self.cidx, 2.0 * np.asarray(zw)[:, None] * np.ones(self.cidx.shape[1])[None, :])
        return h

    def hessian_diag(self, x, z, zw=None):  # oracle/ref_driver.cpp evalHessianDiag
        if self.kind == "quadratic":
            h = self.q.copy()
        elif self.kind == "convex":
            d = 1e-3 + x
            h = 2.0 * self.b * self.b / (d * d * d)
        else:
            h = np.full(len(x), 2.0 * z[0])
            r = x[1:] - x[:-1] ** 2
            h[:-1] += 2.0 - 400.0 * r + 800.0 * x[:-1] ** 2
            h[1:] += 200.0
        return h + self._chain_hessian_diag(zw) if self.chain else h

    def hvec_product(self, x, z, px, zw=None):  # oracle/ref_driver.cpp evalHvecProduct
        if self.kind in ("quadratic", "convex"):
            return self.hessian_diag(x, z, zw) * px
        h = 2.0 * z[0] * px
        r = x[1:] - x[:-1] ** 2
        h[:-1] += (2.0 - 400.0 * r + 800.0 * x[:-1] ** 2) * px[:-1] - 400.0 * x[:-1] * px[1:]
        h[1:] += -400.0 * x[:-1] * px[:-1] + 200.0 * px[1:]
        if self.chain:
            h += self._chain_hessian_diag(zw) * px
        return h

    def eval_obj_con_gradient(self, x):
        if self.chain:  # evalSparseObjConGradient stores the Jacobian entries (.cpp:739-742)
            self._jac = -2.0 * x[self.cidx]
        if self.kind == "quadratic":
            return 0, self.q * x + self.b, [a.copy() for a in self.A]
        if self.kind == "convex":
            d = 1e-3 + x
            return 0, -(self.b * self.b) / (d * d), [-a for a in self.A]
        g = np.zeros_like(x)
        g[:-1] += -2.0 * (1.0 - x[:-1]) + 200.0 * (x[1:] - x[:-1] ** 2) * (-2.0 * x[:-1])
        g[1:] += 200.0 * (x[1:] - x[:-1] ** 2)
        a1 = np.zeros_like(x)
        a1[::2] = 1.0
        return 0, g, [-2.0 * x, a1]


# --------------------------------------------------------------------------------------
# Interior point -- src/ParOptInteriorPoint.cpp (dense constraints only, nwcon = 0)
# --------------------------------------------------------------------------------------
DEFAULT_OPTIONS = {  # src/ParOptInteriorPoint.cpp:536-727
    "max_bound_value": 1e20,
    "abs_res_tol": 1e-6,
    "rel_func_tol": 0.0,
    "abs_step_tol": 0.0,
    "init_barrier_param": 0.1,
    "penalty_gamma": 1000.0,
    "penalty_descent_fraction": 0.3,
    "min_rho_penalty_search": 0.0,
    "init_rho_penalty_search": 0.0,
    "armijo_constant": 1e-5,
    "monotone_barrier_fraction": 0.25,
    "monotone_barrier_power": 1.1,
    "rel_bound_barrier": 1.0,
    "min_fraction_to_boundary": 0.95,
    "qn_sigma": 0.0,
    "function_precision": 1e-10,
    "design_precision": 1e-14,
    "start_affine_multiplier_min": 1.0,
    "use_line_search": True,
    "use_backtracking_alpha": False,
    "sequential_linear_method": False,
    "use_quasi_newton_update": True,
    "qn_subspace_size": 10,
    "max_major_iters": 5000,
    "max_line_iters": 10,
    "iterative_refinement_steps": 1,
    "hessian_reset_freq": 1000000,
    "qn_type": "bfgs",
    "qn_update_type": "skip_negative_curvature",
    "qn_diag_type": "yty_over_yts",
    "norm_type": "infinity",
    "barrier_strategy": "monotone",
    "starting_point_strategy": "affine_step",
    "use_hvec_product": False,
    "use_diag_hessian": False,
    "use_qn_gmres_precon": True,
    "nk_switch_tol": 1e-3,
    "eisenstat_walker_alpha": 1.5,
    "eisenstat_walker_gamma": 1.0,
    "max_gmres_rtol": 0.1,
    "gmres_atol": 1e-30,
    "gmres_subspace_size": 0,
}

LS_SUCCESS, LS_FAILURE, LS_MIN_STEP, LS_MAX_ITERS, LS_NO_IMPROVEMENT, LS_SHORT_STEP = (
    1,
    2,
    4,
    8,
    16,
    32,
)  # src/ParOptInteriorPoint.h:220-225


class Vars:
    """ParOptVars for nwcon = 0: src/ParOptInteriorPoint.h:373-389."""

    def __init__(self, n, c, w=0):
        self.x = np.zeros(n)
        self.zl = np.zeros(n)
        self.zu = np.zeros(n)
        self.z = np.zeros(c)
        self.s = np.zeros(c)
        self.t = np.zeros(c)
        self.zs = np.zeros(c)
        self.zt = np.zeros(c)
        self.zw = np.zeros(w)
        self.sw = np.zeros(w)
        self.tw = np.zeros(w)
        self.zsw = np.zeros(w)
        self.ztw = np.zeros(w)

    NAMES = ("x", "zl", "zu", "z", "s", "t", "zs", "zt", "zw", "sw", "tw", "zsw", "ztw")

    def add(self, o, sign=1.0):  # :128-167
        for k in self.NAMES:
            getattr(self, k)[...] += sign * getattr(o, k)


class InteriorPoint:
    def __init__(self, prob, options=None, comm=None):
        self.prob = prob
        self.comm = comm if comm is not None else prob.comm
        self.ops = VecOps(self.comm)
        self.opt = dict(DEFAULT_OPTIONS)
        if options:
            for k, v in options.items():
                if k not in self.opt:
                    raise KeyError("unknown option %s" % k)
                self.opt[k] = v
        o = self.opt
        n, c = prob.nlocal, prob.c
        self.n, self.c = n, c
        w = getattr(prob, "nwcon", 0)
        self.w = w
        self.ninequality = c
        self.use_lower = bool(getattr(prob, "use_lower", True))
        self.use_upper = bool(getattr(prob, "use_upper", True))
        self.vars = Vars(n, c, w)
        self.res = Vars(n, c, w)
        self.step = Vars(n, c, w)
        self.refine = Vars(n, c, w)
        qt = o["qn_type"]
        if qt == "bfgs":
            self.qn = LBFGS(n, o["qn_subspace_size"], self.ops, o["qn_update_type"])
        elif qt == "sr1":
            self.qn = LSR1(n, o["qn_subspace_size"], self.ops)
        else:
            self.qn = None
        if self.qn is not None:
            dt = o["qn_diag_type"]
            self.qn.diag_type = "yts_over_sts" if dt in ("yts_over_sts",) else "yty_over_yts"
        gamma = o["penalty_gamma"]
        self.gamma_s = np.array([0.0 if i < self.ninequality else gamma for i in range(c)])
        self.gamma_t = np.full(c, gamma)
        nwineq = getattr(prob, "nwineq", 0)
        self.gamma_sw = np.array([0.0 if i < nwineq else gamma for i in range(w)])  # :361-374
        self.gamma_tw = np.full(w, gamma)
        self.Cw = np.zeros(w)
        self.barrier_param = o["init_barrier_param"]
        self.rho = o["init_rho_penalty_search"]
        self.niter = self.neval = self.ngeval = 0
        self.fobj = 0.0
        self.cvals = np.zeros(c)
        self.g = np.zeros(n)
        self.Ac = [np.zeros(n) for _ in range(c)]
        self.Dinv = np.ones(n)
        self.Glu = None
        self.Celu = None
        self.hdiag = np.zeros(n) if o["use_diag_hessian"] else None  # :293-297 (zero until evaluated)
        self.nhvec = 0
        self.trace = []
        self.hook = None  # called as hook(self, k) at the top of every iteration
        self._init_and_check_bounds()
        v = self.vars
        v.zl[:] = 1.0
        v.zu[:] = 1.0
        v.z[:] = 1.0
        v.s[:] = 1.0
        v.t[:] = 1.0
        v.zs[:] = 1.0
        v.zt[:] = 1.0
        for k in ("zw", "sw", "tw", "zsw", "ztw"):
            getattr(v, k)[:] = 1.0

    # ---- driver-facing setters (used by the trust-region driver) -------------------------
    def set_quasi_newton(self, qn):  # :1193-1234
        self.qn = qn

    def reset_problem_instance(self, prob):  # :745-764 (sizes must match)
        assert (prob.nlocal, prob.c, getattr(prob, "nwcon", 0)) == (self.n, self.c, self.w)
        self.prob = prob

    def reset_design_and_bounds(self):  # :1249-1251
        x, lb, ub = self.prob.vars_and_bounds()
        self.vars.x[:] = x
        self.lb, self.ub = lb.copy(), ub.copy()

    def set_penalty_gamma(self, gamma):  # scalar :1128-1151, array :1160-1172 (dense blocks only)
        if np.isscalar(gamma):
            if gamma >= 0.0:
                nwineq = getattr(self.prob, "nwineq", 0)
                self.gamma_s = np.array([0.0 if i < self.ninequality else gamma for i in range(self.c)])
                self.gamma_t = np.full(self.c, float(gamma))
                self.gamma_sw = np.array([0.0 if i < nwineq else gamma for i in range(self.w)])
                self.gamma_tw = np.full(self.w, float(gamma))
        else:
            for i in range(self.c):
                if gamma[i] >= 0.0:
                    self.gamma_s[i] = 0.0 if i < self.ninequality else gamma[i]
                    self.gamma_t[i] = gamma[i]

    def _csr_form(self):
        """The problem is a ParOptSparseProblem (general sparse S): the built-in chain constraints, or any
        problem object that sets csr_form = True and provides sparse_jacobian_dense()."""
        return bool(getattr(self.prob, "chain", None)) or bool(getattr(self.prob, "csr_form", False))

    # ---- quasi-definite block matrix (nwblock = 1): src/ParOptSparseMat.cpp:41-229 --------
    def _factor(self, v, Cdiag):
        """Cw = 1 / (Cdiag + Aw Dinv Aw^T) per constraint."""
        if self.w and self._csr_form():
            # ParOptQuasiDefSparseMat::factor (src/ParOptSparseMat.cpp:303-356): S = C + Aw D^-1 Aw^T, here
            # dense (the factorization is a direct solve; its sparsity is an implementation matter)
            Aw = self.prob.sparse_jacobian_dense()
            self._Aw = Aw
            self._Schol = sla.cho_factor(np.diag(Cdiag) + (Aw * self.Dinv) @ Aw.T, lower=True, check_finite=False)
        elif self.w:
            A = Cdiag.copy()
            self.prob.add_sparse_inner_product(1.0, self.Dinv, A)
            self.Cw = 1.0 / A

    def _apply(self, bx, bw=None):
        """[D Aw^T; Aw -C] [yx; -yw] = [bx; bw]  ->  (yx, yw)   (:122-190)."""
        yx = self.Dinv * bx
        if self.w == 0:
            return yx, np.zeros(0)
        yw = np.zeros(self.w) if bw is None else bw.copy()
        self.prob.add_sparse_jacobian(-1.0, yx, yw)
        if self._csr_form():  # :358-431
            yw = sla.cho_solve(self._Schol, yw, check_finite=False)
        else:
            yw = yw * self.Cw
        yx = bx.copy()
        self.prob.add_sparse_jacobian_transpose(1.0, yw, yx)
        yx = yx * self.Dinv
        return yx, yw

    # ---- bound masks -----------------------------------------------------------------
    def _masks(self):
        mb = self.opt["max_bound_value"]
        return (self.lb > -mb) & self.use_lower, (self.ub < mb) & self.use_upper

    def _init_and_check_bounds(self):  # :4277-4361
        mb = self.opt["max_bound_value"]
        x, lb, ub = self.prob.vars_and_bounds()
        x, lb, ub = x.copy(), lb.copy(), ub.copy()
        rel_bound = 0.001 * self.barrier_param
        flag = 0
        if self.use_lower and self.use_upper:
            L = lb > -mb
            U = ub < mb
            both = L & U
            bad = both & (lb >= ub)
            if np.any(bad):
                flag |= 1
                lbn = 0.5 * (lb[bad] + ub[bad]) - 0.5 * rel_bound
                lb[bad] = lbn
                ub[bad] = lbn + rel_bound
            delta = np.where(both, ub - lb, 1.0)
            lo = L & (x < lb + rel_bound * delta)
            if np.any(lo):
                flag |= 2
            x = np.where(lo, lb + rel_bound * delta, x)
            hi = U & (x > ub - rel_bound * delta)
            if np.any(hi):
                flag |= 4
            x = np.where(hi, ub - rel_bound * delta, x)
        # MPI_Allreduce(MPI_BOR) of the reference (:4326-4327), accumulated over the calls like the warnings are
        fl = np.array([float(flag & 1), float(flag & 2), float(flag & 4)])
        fl = self.comm.allreduce(fl, op="max") if hasattr(self.comm, "allreduce") else fl
        self.check_flag = getattr(self, "check_flag", 0) | (1 if fl[0] else 0) | (2 if fl[1] else 0) | (4 if fl[2] else 0)
        self.vars.x[:] = x
        self.lb, self.ub = lb, ub
        self.vars.zl[lb <= -mb] = 0.0
        self.vars.zu[ub >= mb] = 0.0

    # ---- residuals -------------------------------------------------------------------
    def compute_kkt_res(self, v, barrier, r):  # :1337-1446
        o = self.opt
        beta = o["rel_bound_barrier"]
        L, U = self._masks()
        rx = v.zl.copy() if self.use_lower else np.zeros(self.n)
        if self.use_upper:
            rx += -1.0 * v.zu
        rx += -1.0 * self.g
        for i in range(self.c):
            rx += v.z[i] * self.Ac[i]
        if self.w:  # :1358-1398
            self.prob.add_sparse_jacobian_transpose(1.0, v.zw, rx)
            r.zw[:] = -(self.prob.eval_sparse_con(v.x) - v.sw + v.tw)
            r.sw[:] = v.zsw - self.gamma_sw - v.zw
            r.tw[:] = v.ztw - self.gamma_tw + v.zw
            r.zsw[:] = barrier - v.sw * v.zsw
            r.ztw[:] = barrier - v.tw * v.ztw
        r.x[:] = rx
        r.z[:] = -(self.cvals - v.s + v.t)
        r.s[:] = -(self.gamma_s - v.zs + v.z)
        r.t[:] = -(self.gamma_t - v.zt - v.z)
        r.zs[:] = -(v.s * v.zs - barrier)
        r.zt[:] = -(v.t * v.zt - barrier)
        r.zl[:] = np.where(L, -((v.x - self.lb) * v.zl - beta * barrier), 0.0)
        r.zu[:] = np.where(U, -((self.ub - v.x) * v.zu - beta * barrier), 0.0)

    def add_kkt_res_step(self, v, p, r, inexact_newton_step=0):  # :1451-1583
        o = self.opt
        L, U = self._masks()
        if inexact_newton_step:
            r.x += -1.0 * self.prob.hvec_product(v.x, v.z, p.x, v.zw if self.w else None)
        elif o["use_diag_hessian"]:
            r.x -= p.x * self.hdiag
        else:
            if self.qn is not None and not o["sequential_linear_method"]:
                self.qn.mult_add(-1.0, p.x, r.x)
            if o["qn_sigma"] != 0.0:
                r.x += -o["qn_sigma"] * p.x
        for i in range(self.c):
            r.x += p.z[i] * self.Ac[i]
        if self.use_lower:
            r.x += p.zl
        if self.use_upper:
            r.x += -1.0 * p.zu
        if self.w:  # :1492-1527
            self.prob.add_sparse_jacobian_transpose(1.0, p.zw, r.x)
            self.prob.add_sparse_jacobian(-1.0, p.x, r.zw)
            r.zw += p.sw
            r.zw += -1.0 * p.tw
            r.sw += p.zsw
            r.sw += -1.0 * p.zw
            r.tw += p.ztw
            r.tw += p.zw
            r.zsw -= p.sw * v.zsw + v.sw * p.zsw
            r.ztw -= p.tw * v.ztw + v.tw * p.ztw
        for i in range(self.c):
            r.z[i] -= self.ops.dot(self.Ac[i], p.x) - p.s[i] + p.t[i]
            r.s[i] += p.zs[i] - p.z[i]
            r.t[i] += p.zt[i] + p.z[i]
            r.zs[i] -= p.s[i] * v.zs[i] + v.s[i] * p.zs[i]
            r.zt[i] -= p.t[i] * v.zt[i] + v.t[i] * p.zt[i]
        r.zl[:] -= np.where(L, (v.x - self.lb) * p.zl + p.x * v.zl, 0.0)
        r.zu[:] -= np.where(U, (self.ub - v.x) * p.zu - p.x * v.zu, 0.0)

    def add_mehrotra_corrector_residual(self, p, r):  # :1729-1789
        L, U = self._masks()
        r.zs[:] -= p.s * p.zs
        r.zt[:] -= p.t * p.zt
        r.zsw[:] -= p.sw * p.zsw
        r.ztw[:] -= p.tw * p.ztw
        r.zl[:] -= np.where(L, p.x * p.zl, 0.0)
        r.zu[:] += np.where(U, p.x * p.zu, 0.0)

    def compute_res_norm(self, r):  # :1588-1723
        nt = self.opt["norm_type"]
        ops = self.ops
        if nt == "infinity":
            mp = ops.maxabs(r.x)
            mi = ops.maxabs(r.zw)
            md = max(ops.maxabs(r.sw), ops.maxabs(r.tw), ops.maxabs(r.zsw), ops.maxabs(r.ztw))
            for i in range(self.c):
                mp = max(mp, abs(r.s[i]), abs(r.t[i]))
                mi = max(mi, abs(r.z[i]))
                md = max(md, abs(r.zs[i]), abs(r.zt[i]))
            if self.use_lower:
                md = max(md, ops.maxabs(r.zl))
            if self.use_upper:
                md = max(md, ops.maxabs(r.zu))
        elif nt == "l1":
            mp = ops.l1norm(r.x) + float(np.sum(np.abs(r.s)) + np.sum(np.abs(r.t)))
            mi = ops.l1norm(r.zw) + float(np.sum(np.abs(r.z)))
            md = ops.l1norm(r.sw) + ops.l1norm(r.tw) + ops.l1norm(r.zsw) + ops.l1norm(r.ztw)
            md += float(np.sum(np.abs(r.zs)) + np.sum(np.abs(r.zt)))
            md += ops.l1norm(r.zl) + ops.l1norm(r.zu)
        else:
            mp = ops.norm(r.x) ** 2 + float(np.sum(r.s**2 + r.t**2))
            mi = ops.norm(r.zw) ** 2 + float(np.sum(r.z**2))
            # the reference squares l1 norms of the sparse dual blocks here (:1633-1638)
            md = ops.l1norm(r.sw) ** 2 + ops.l1norm(r.tw) ** 2 + ops.l1norm(r.zsw) ** 2 + ops.l1norm(r.ztw) ** 2
            md += float(np.sum(r.zs**2 + r.zt**2)) + ops.norm(r.zl) ** 2 + ops.norm(r.zu) ** 2
            mp, mi, md = math.sqrt(mp), math.sqrt(mi), math.sqrt(md)
        return mp, md, mi, max(mp, md, mi)

    # ---- KKT system ------------------------------------------------------------------
    def setup_kkt_diag_system(self, v, use_qn):  # :1832-1971
        o = self.opt
        L, U = self._masks()
        b0 = 0.0
        if o["use_diag_hessian"] and self.hdiag is not None:  # :1840-1842
            b0 = self.hdiag
        elif self.qn is not None and use_qn:
            b0 = self.qn.b0
        d = np.zeros(self.n) + b0 + o["qn_sigma"]
        d = d + np.where(L, v.zl / np.where(L, v.x - self.lb, 1.0), 0.0)
        d = d + np.where(U, v.zu / np.where(U, self.ub - v.x, 1.0), 0.0)
        self.Dinv = 1.0 / d
        self._factor(v, v.sw / v.zsw + v.tw / v.ztw if self.w else None)  # Cdiag :1912-1930
        c = self.c
        G = np.zeros((c, c))
        for j in range(c):
            xt, _ = self._apply(self.Ac[j])
            for i in range(j, c):
                G[i, j] += self.ops.dot(self.Ac[i], xt)
        for j in range(c):
            for i in range(j + 1, c):
                G[j, i] = G[i, j]
        for i in range(c):
            G[i, i] += v.s[i] / v.zs[i] + v.t[i] / v.zt[i]
        self.G = G.copy()
        self.Glu = sla.lu_factor(G, check_finite=False) if c > 0 else None

    def _gsolve(self, rhs):
        if self.c == 0:
            return rhs
        return sla.lu_solve(self.Glu, rhs, check_finite=False)

    def solve_kkt_diag_full(self, v, b, y):  # :2074-2243
        L, U = self._masks()
        xl = np.where(L, v.x - self.lb, 1.0)
        xu = np.where(U, self.ub - v.x, 1.0)
        d1 = b.x.copy()
        d1 += np.where(L, b.zl / xl, 0.0)
        d1 -= np.where(U, b.zu / xu, 0.0)
        d2 = None
        if self.w:  # :2111-2136
            d2 = b.zw + (b.zsw + v.sw * b.sw) / v.zsw - (b.ztw + v.tw * b.tw) / v.ztw
        yx, _ = self._apply(d1, d2)
        yz = self.ops.mdot(yx, self.Ac)
        yz = b.z + (b.zs + v.s * b.s) / v.zs - (b.zt + v.t * b.t) / v.zt - yz
        yz = self._gsolve(yz)
        y.z[:] = yz
        y.zs[:] = yz - b.s
        y.zt[:] = -b.t - yz
        y.s[:] = (b.zs - v.s * y.zs) / v.zs
        y.t[:] = (b.zt - v.t * y.zt) / v.zt
        for i in range(self.c):
            d1 += yz[i] * self.Ac[i]
        yx, yw = self._apply(d1, d2)
        y.x[:] = yx
        if self.w:  # :2180-2208
            y.zw[:] = yw
            y.zsw[:] = yw - b.sw
            y.ztw[:] = -b.tw - yw
            y.sw[:] = (b.zsw - v.sw * y.zsw) / v.zsw
            y.tw[:] = (b.ztw - v.tw * y.ztw) / v.ztw
        y.zl[:] = np.where(L, (b.zl - v.zl * y.x) / xl, 0.0)
        y.zu[:] = np.where(U, (b.zu + v.zu * y.x) / xu, 0.0)

    def solve_kkt_diag_bx(self, v, bx, y):  # :2257-2369
        L, U = self._masks()
        xl = np.where(L, v.x - self.lb, 1.0)
        xu = np.where(U, self.ub - v.x, 1.0)
        d1 = bx.copy()
        d2 = np.zeros(self.w) if self.w else None
        yx, _ = self._apply(d1, d2)
        yz = -self.ops.mdot(yx, self.Ac)
        yz = self._gsolve(yz)
        y.z[:] = yz
        y.zs[:] = yz
        y.zt[:] = -yz
        y.s[:] = -(v.s * y.zs) / v.zs
        y.t[:] = -(v.t * y.zt) / v.zt
        for i in range(self.c):
            d1 += yz[i] * self.Ac[i]
        yx, yw = self._apply(d1, d2)
        y.x[:] = yx
        if self.w:  # :2312-2334
            y.zw[:] = yw
            y.zsw[:] = yw
            y.ztw[:] = -yw
            y.sw[:] = -(v.sw * y.zsw) / v.zsw
            y.tw[:] = -(v.tw * y.ztw) / v.ztw
        y.zl[:] = np.where(L, -(v.zl * y.x) / xl, 0.0)
        y.zu[:] = np.where(U, (v.zu * y.x) / xu, 0.0)

    def solve_kkt_diag_x(self, v, bx):  # :2385-2428 -> yx only
        d1 = bx.copy()
        yx, _ = self._apply(d1)
        yz = -self.ops.mdot(yx, self.Ac)
        yz = self._gsolve(yz)
        for i in range(self.c):
            d1 += yz[i] * self.Ac[i]
        return self._apply(d1)[0]

    def solve_kkt_diag_alpha(self, v, bx, alpha, b, y):  # :2441-2614
        L, U = self._masks()
        xl = np.where(L, v.x - self.lb, 1.0)
        xu = np.where(U, self.ub - v.x, 1.0)
        d1 = bx.copy()
        d1 += alpha * np.where(L, b.zl / xl, 0.0)
        d1 -= alpha * np.where(U, b.zu / xu, 0.0)
        d2 = None
        if self.w:  # :2482-2508
            d2 = alpha * (b.zw + (b.zsw + v.sw * b.sw) / v.zsw - (b.ztw + v.tw * b.tw) / v.ztw)
        yx, _ = self._apply(d1, d2)
        yz = self.ops.mdot(yx, self.Ac)
        yz = alpha * (b.z + (b.zs + v.s * b.s) / v.zs - (b.zt + v.t * b.t) / v.zt) - yz
        yz = self._gsolve(yz)
        y.z[:] = yz
        y.zs[:] = yz - alpha * b.s
        y.zt[:] = -alpha * b.t - yz
        y.s[:] = (alpha * b.zs - v.s * y.zs) / v.zs
        y.t[:] = (alpha * b.zt - v.t * y.zt) / v.zt
        for i in range(self.c):
            d1 += yz[i] * self.Ac[i]
        yx, yw = self._apply(d1, d2)
        y.x[:] = yx
        if self.w:  # :2557-2585
            y.zw[:] = yw
            y.zsw[:] = yw - alpha * b.sw
            y.ztw[:] = -alpha * b.tw - yw
            y.sw[:] = (alpha * b.zsw - v.sw * y.zsw) / v.zsw
            y.tw[:] = (alpha * b.ztw - v.tw * y.ztw) / v.ztw
        y.zl[:] = np.where(L, (alpha * b.zl - v.zl * y.x) / xl, 0.0)
        y.zu[:] = np.where(U, (alpha * b.zu + v.zu * y.x) / xu, 0.0)

    def eval_obj_barrier_deriv(self, v, p):  # :5669-5766
        beta = self.opt["rel_bound_barrier"]
        _, _, L, U, dl, du = self._barrier_sums(v.x)
        px = p.x
        ql = np.where(L, px / dl, 0.0)
        qu = np.where(U, px / du, 0.0)
        ppos = beta * float(np.sum(np.where(L & (px > 0.0), ql, 0.0)) - np.sum(np.where(U & ~(px > 0.0), qu, 0.0)))
        pneg = beta * float(np.sum(np.where(L & ~(px > 0.0), ql, 0.0)) - np.sum(np.where(U & (px > 0.0), qu, 0.0)))
        if self.w:  # :5711-5730
            for val, pv in ((v.sw, p.sw), (v.tw, p.tw)):
                q = pv / val
                ppos += float(np.sum(q[pv > 0.0]))
                pneg += float(np.sum(q[~(pv > 0.0)]))
        out = self.comm.allreduce([ppos, pneg])
        ppos, pneg = float(out[0]), float(out[1])
        for i in range(self.c):
            for val, pv in ((v.s[i], p.s[i]), (v.t[i], p.t[i])):
                if pv > 0.0:
                    ppos += pv / val
                else:
                    pneg += pv / val
        pmerit = self.ops.dot(self.g, p.x) - self.barrier_param * (ppos + pneg)
        for i in range(self.c):
            pmerit += self.gamma_s[i] * p.s[i] + self.gamma_t[i] * p.t[i]
        if self.w:  # :5767
            pmerit += self.ops.dot(self.gamma_sw, p.sw) + self.ops.dot(self.gamma_tw, p.tw)
        return pmerit

    def _smw_correct(self, v, p, scratch, use_qn):
        """p -= K0^-1-solve of Z Ce^-1 Z^T p.x  (the second half of computeKKTStep, :2718-2735)."""
        Z = self.qn.get_compact()[3] if (self.qn is not None and use_qn) else []
        if len(Z) > 0:
            zt = self.ops.mdot(p.x, Z)
            zt = sla.lu_solve(self.Celu, zt, check_finite=False)
            xt = np.zeros(self.n)
            for i in range(len(Z)):
                xt += zt[i] * Z[i]
            self.solve_kkt_diag_bx(v, xt, scratch)
            return scratch
        return None

    def compute_kkt_gmres_step(self, v, r, p, rtol, atol, use_qn):  # :5796-6191
        m = self.opt["gmres_subspace_size"]
        if m <= 0:
            return 0
        c = self.c
        H = np.zeros((m + 1) * (m + 2) // 2)
        alpha = np.zeros(m + 1)
        gres = np.zeros(m + 1)
        y = np.zeros(m)
        fproj = np.zeros(m)
        aproj = np.zeros(m)
        Qcos, Qsin = np.zeros(m), np.zeros(m)
        W = [None] * (m + 1)
        beta = float(np.sum(r.z**2 + r.s**2 + r.t**2 + r.zs**2 + r.zt**2))
        if self.use_lower:
            beta += self.ops.dot(r.zl, r.zl)
        if self.use_upper:
            beta += self.ops.dot(r.zu, r.zu)
        if self.w:  # :5846-5852
            for k in ("zw", "sw", "tw", "zsw", "ztw"):
                beta += self.ops.dot(getattr(r, k), getattr(r, k))
        bnorm = math.sqrt(self.ops.dot(r.x, r.x) + beta)
        beta *= 1.0 / (bnorm * bnorm)
        cinfeas = float(np.sum((self.cvals - v.s + v.t) ** 2))
        cscale = 0.0
        if cinfeas != 0.0:
            cinfeas = math.sqrt(cinfeas)
            cscale = 1.0 / cinfeas
        cwinfeas, cwscale = 0.0, 0.0
        if self.w:  # :5884-5890
            cwinfeas = math.sqrt(self.ops.dot(r.zw, r.zw))
            if cwinfeas != 0.0:
                cwscale = 1.0 / cwinfeas
        awproj = np.zeros(m)
        gres[0] = bnorm
        W[0] = r.x / gres[0]
        alpha[0] = 1.0
        niters = 0
        scratch = Vars(self.n, c, self.w)
        for i in range(m):
            self.solve_kkt_diag_alpha(v, W[i], alpha[i] / bnorm, r, p)
            corr = self._smw_correct(v, p, scratch, use_qn)
            if corr is not None:
                p.x += -1.0 * corr.x  # only the design part is corrected inside the loop (:5949)
            fproj[i] = self.eval_obj_barrier_deriv(v, p)
            aproj[i] = 0.0
            for j in range(c):
                cj = self.ops.dot(self.Ac[j], p.x) - p.s[j] + p.t[j]
                aproj[i] -= cscale * r.z[j] * cj
            if self.w:  # :5963-5973
                xt = np.zeros(self.n)
                self.prob.add_sparse_jacobian_transpose(1.0, r.zw, xt)
                awproj[i] = -cwscale * self.ops.dot(p.x, xt)
                awproj[i] += cwscale * self.ops.dot(r.zw, p.sw)
                awproj[i] -= cwscale * self.ops.dot(r.zw, p.tw)
            W[i + 1] = self.prob.hvec_product(v.x, v.z, p.x, v.zw if self.w else None)
            self.nhvec += 1
            if self.qn is not None and use_qn:
                self.qn.mult_add(-1.0, p.x, W[i + 1])
            W[i + 1] += 1.0 * W[i]
            alpha[i + 1] = alpha[i]
            hptr = (i + 1) * (i + 2) // 2 - 1
            for j in range(i, -1, -1):
                H[j + hptr] = self.ops.dot(W[i + 1], W[j]) + beta * alpha[i + 1] * alpha[j]
                W[i + 1] += -H[j + hptr] * W[j]
                alpha[i + 1] -= H[j + hptr] * alpha[j]
            H[i + 1 + hptr] = math.sqrt(self.ops.dot(W[i + 1], W[i + 1]) + beta * alpha[i + 1] * alpha[i + 1])
            W[i + 1] *= 1.0 / H[i + 1 + hptr]
            alpha[i + 1] *= 1.0 / H[i + 1 + hptr]
            for k in range(i):
                h1, h2 = H[k + hptr], H[k + 1 + hptr]
                H[k + hptr] = h1 * Qcos[k] + h2 * Qsin[k]
                H[k + 1 + hptr] = -h1 * Qsin[k] + h2 * Qcos[k]
            h1, h2 = H[i + hptr], H[i + 1 + hptr]
            sq = math.sqrt(h1 * h1 + h2 * h2)
            Qcos[i], Qsin[i] = h1 / sq, h2 / sq
            H[i + hptr] = h1 * Qcos[i] + h2 * Qsin[i]
            H[i + 1 + hptr] = -h1 * Qsin[i] + h2 * Qcos[i]
            h1 = gres[i]
            gres[i] = h1 * Qcos[i]
            gres[i + 1] = -h1 * Qsin[i]
            niters += 1
            for j in range(niters - 1, -1, -1):
                y[j] = gres[j]
                for k in range(j + 1, niters):
                    hp = (k + 1) * (k + 2) // 2 - 1
                    y[j] = y[j] - H[j + hp] * y[k]
                hp = (j + 1) * (j + 2) // 2 - 1
                y[j] = y[j] / H[j + hp]
            fpr = float(np.dot(y[:niters], fproj[:niters]))
            cpr = float(np.dot(y[:niters], aproj[:niters] + awproj[:niters]))
            constraint_descent = 1 if cpr <= -0.01 * (cinfeas + cwinfeas) else 0
            if fpr < 0.0 or constraint_descent:
                if abs(gres[i + 1]) < atol or abs(gres[i + 1]) < rtol * bnorm:
                    break
        for i in range(niters - 1, -1, -1):
            for j in range(i + 1, niters):
                hp = (j + 1) * (j + 2) // 2 - 1
                gres[i] = gres[i] - H[i + hp] * gres[j]
            hp = (i + 1) * (i + 2) // 2 - 1
            gres[i] = gres[i] / H[i + hp]
        W[0] = W[0] * gres[0]
        gamma = gres[0] * alpha[0]
        for i in range(1, niters):
            W[0] += gres[i] * W[i]
            gamma += gres[i] * alpha[i]
        gamma /= bnorm
        r.x[:] = W[0]
        for k in ("z", "s", "t", "zs", "zt", "zl", "zu") + (("zw", "sw", "tw", "zsw", "ztw") if self.w else ()):
            getattr(r, k)[...] *= gamma
        self.solve_kkt_diag_full(v, r, p)
        corr = self._smw_correct(v, p, scratch, use_qn)
        if corr is not None:
            p.add(corr, -1.0)
        fpr = self.eval_obj_barrier_deriv(v, p)
        cpr = 0.0
        for i in range(c):
            deriv = self.ops.dot(self.Ac[i], p.x) - p.s[i] + p.t[i]
            cpr += cscale * (self.cvals[i] - v.s[i] + v.t[i]) * deriv
        if self.w:  # :6163-6174 (both slack terms are SUBTRACTED here, unlike the loop above)
            rzw = self.prob.eval_sparse_con(v.x) - v.sw + v.tw
            xt = np.zeros(self.n)
            self.prob.add_sparse_jacobian_transpose(1.0, rzw, xt)
            cpr += cwscale * self.ops.dot(p.x, xt)
            cpr -= cwscale * self.ops.dot(p.sw, rzw)
            cpr -= cwscale * self.ops.dot(p.tw, rzw)
        if fpr < 0.0 or cpr < -0.01 * (cinfeas + cwinfeas):
            return niters
        return -niters

    def setup_kkt_system(self, v, use_qn):  # :2634-2667
        self.Celu = None
        if self.qn is not None and use_qn:
            b0, d0, M, Z = self.qn.get_compact()
            k = len(Z)
            if k > 0:
                Ce = np.zeros((k, k))
                for i in range(k):
                    xt = self.solve_kkt_diag_x(v, Z[i])
                    Ce[:, i] = self.ops.mdot(xt, Z)
                Ce -= M / np.outer(d0, d0)
                self.Ce = Ce.copy()
                self.Celu = sla.lu_factor(Ce, check_finite=False)

    def compute_kkt_step(self, v, r, p, use_qn):  # :2700-2737 (r is clobbered)
        Z = []
        if self.qn is not None and use_qn:
            Z = self.qn.get_compact()[3]
        self.solve_kkt_diag_full(v, r, p)
        if len(Z) > 0:
            zt = self.ops.mdot(p.x, Z)
            zt = sla.lu_solve(self.Celu, zt, check_finite=False)
            xt = np.zeros(self.n)
            for i in range(len(Z)):
                xt += zt[i] * Z[i]
            self.solve_kkt_diag_bx(v, xt, r)
            p.add(r, -1.0)

    # ---- complementarity / step lengths ---------------------------------------------
    def compute_comp(self, v):  # :2742-2820
        L, U = self._masks()
        prod = float(np.sum(np.where(L, v.zl * (v.x - self.lb), 0.0)))
        prod += float(np.sum(np.where(U, v.zu * (self.ub - v.x), 0.0)))
        cnt = float(np.count_nonzero(L) + np.count_nonzero(U))
        prod = prod / self.opt["rel_bound_barrier"]
        prod += float(np.sum(v.sw * v.zsw + v.tw * v.ztw))
        cnt += 2.0 * self.w
        out = self.comm.allreduce([prod, cnt])
        prod, cnt = float(out[0]), float(out[1])
        prod += float(np.sum(v.s * v.zs + v.t * v.zt))
        cnt += 2.0 * self.c
        return prod / cnt if cnt != 0.0 else 0.0

    def compute_comp_step(self, v, ax, az, p):  # :2825-2923
        L, U = self._masks()
        xn = v.x + ax * p.x
        prod = float(np.sum(np.where(L, (v.zl + az * p.zl) * (xn - self.lb), 0.0)))
        prod += float(np.sum(np.where(U, (v.zu + az * p.zu) * (self.ub - xn), 0.0)))
        cnt = float(np.count_nonzero(L) + np.count_nonzero(U))
        prod = prod / self.opt["rel_bound_barrier"]
        prod += float(np.sum((v.sw + ax * p.sw) * (v.zsw + az * p.zsw) + (v.tw + ax * p.tw) * (v.ztw + az * p.ztw)))
        cnt += 2.0 * self.w
        out = self.comm.allreduce([prod, cnt])
        prod, cnt = float(out[0]), float(out[1])
        prod += float(
            np.sum((v.s + ax * p.s) * (v.zs + az * p.zs) + (v.t + ax * p.t) * (v.zt + az * p.zt))
        )
        cnt += 2.0 * self.c
        return prod / cnt if cnt != 0.0 else 0.0

    @staticmethod
    def _min_ratio(tau, num, den, mask):
        if not np.any(mask):
            return 1.0
        return float(min(1.0, np.min(-tau * num[mask] / den[mask])))

    def compute_max_step(self, v, tau, p):  # :2942-3103 (primal loops are NOT masked)
        mx, mz = 1.0, 1.0
        if self.use_lower:
            m = p.x < 0.0
            mx = min(mx, self._min_ratio(tau, v.x - self.lb, p.x, m))
        if self.use_upper:
            m = p.x > 0.0
            if np.any(m):
                mx = min(mx, float(np.min(tau * (self.ub - v.x)[m] / p.x[m])))
        mx = min(mx, self._min_ratio(tau, v.s, p.s, p.s < 0.0))
        mx = min(mx, self._min_ratio(tau, v.t, p.t, p.t < 0.0))
        mz = min(mz, self._min_ratio(tau, v.zs, p.zs, p.zs < 0.0))
        mz = min(mz, self._min_ratio(tau, v.zt, p.zt, p.zt < 0.0))
        mx = min(mx, self._min_ratio(tau, v.sw, p.sw, p.sw < 0.0))  # :3017-3061
        mx = min(mx, self._min_ratio(tau, v.tw, p.tw, p.tw < 0.0))
        mz = min(mz, self._min_ratio(tau, v.zsw, p.zsw, p.zsw < 0.0))
        mz = min(mz, self._min_ratio(tau, v.ztw, p.ztw, p.ztw < 0.0))
        if self.use_lower:
            mz = min(mz, self._min_ratio(tau, v.zl, p.zl, p.zl < 0.0))
        if self.use_upper:
            mz = min(mz, self._min_ratio(tau, v.zu, p.zu, p.zu < 0.0))
        out = self.comm.allreduce([mx, mz], "min")
        return float(out[0]), float(out[1])

    def scale_kkt_step(self, v, p, tau, comp, inexact_newton_step=0):  # :3196-3274
        ax, az = self.compute_max_step(v, tau, p)
        ceq = 0
        bnd = 100.0
        if inexact_newton_step:  # Newton step: one common step length (:3240-3248)
            if ax > az:
                ax = az
            else:
                az = ax
        else:
            if ax > az:
                if ax > bnd * az:
                    ax = bnd * az
                elif ax < az / bnd:
                    ax = az / bnd
            else:
                if az > bnd * ax:
                    az = bnd * ax
                elif az < ax / bnd:
                    az = ax / bnd
            comp_new = self.compute_comp_step(v, ax, az, p)
            if comp_new > 10.0 * comp:
                ceq = 1
                if ax > az:
                    ax = az
                else:
                    az = ax
        p.x *= ax
        p.zl *= az
        p.zu *= az
        p.sw *= ax
        p.tw *= ax
        p.zw *= az
        p.zsw *= az
        p.ztw *= az
        p.s *= ax
        p.t *= ax
        p.z *= az
        p.zs *= az
        p.zt *= az
        return ceq, ax, az

    # ---- merit function --------------------------------------------------------------
    def _clamp_step(self, x, alpha, p, lb=None, ub=None, lower_value=None):  # computeStep :3146-3191
        eps = self.opt["design_precision"]
        x = x + alpha * p
        if lb is not None:
            x = np.where(x <= lb + eps, lb + eps, x)
        elif lower_value is not None:
            x = np.where(x <= lower_value + eps, lower_value + eps, x)
        if ub is not None:
            x = np.where(x + eps >= ub, ub - eps, x)
        return x

    def _barrier_sums(self, x):
        L, U = self._masks()
        dl = np.where(L, x - self.lb, 1.0)
        du = np.where(U, self.ub - x, 1.0)
        ll = np.log(dl)
        lu = np.log(du)
        pos = float(np.sum(np.where(L & (dl > 1.0), ll, 0.0)) + np.sum(np.where(U & (du > 1.0), lu, 0.0)))
        neg = float(np.sum(np.where(L & ~(dl > 1.0), ll, 0.0)) + np.sum(np.where(U & ~(du > 1.0), lu, 0.0)))
        return pos, neg, L, U, dl, du

    @staticmethod
    def _log_split(vals):
        if vals.size == 0:
            return 0.0, 0.0
        lg = np.log(vals)
        big = vals > 1.0
        return float(np.sum(lg[big])), float(np.sum(lg[~big]))

    def eval_merit_func(self, fk, ck, xk, sk, tk, swk=None, twk=None):  # :3524-3637
        beta = self.opt["rel_bound_barrier"]
        pos, neg, *_ = self._barrier_sums(xk)
        pos, neg = pos * beta, neg * beta
        if swk is None:
            swk = twk = np.zeros(0)
        for arr in (swk, twk):  # :3572-3590 (not scaled by rel_bound_barrier)
            a, b_ = self._log_split(arr)
            pos += a
            neg += b_
        out = self.comm.allreduce([pos, neg])
        pos, neg = float(out[0]), float(out[1])
        for i in range(self.c):
            for val in (sk[i], tk[i]):
                if val > 1.0:
                    pos += math.log(val)
                else:
                    neg += math.log(val)
        dense = float(np.sum((ck - sk + tk) ** 2))
        sparse = 0.0
        if self.w:  # evalInfeas :3438-3462
            rw = self.prob.eval_sparse_con(xk) - swk + twk
            sparse = self.ops.norm(rw)
        infeas = math.sqrt(dense + sparse * sparse)
        merit = (fk + (self.ops.dot(self.gamma_sw, swk) + self.ops.dot(self.gamma_tw, twk))
                 - self.barrier_param * (pos + neg) + self.rho * infeas)
        for i in range(self.c):
            merit += self.gamma_s[i] * sk[i] + self.gamma_t[i] * tk[i]
        return merit

    def eval_merit_init_deriv(self, v, p, max_x):  # :3652-3924
        o = self.opt
        beta = o["rel_bound_barrier"]
        pos, neg, L, U, dl, du = self._barrier_sums(v.x)
        px = p.x
        ql = np.where(L, px / dl, 0.0)
        qu = np.where(U, px / du, 0.0)
        ppos = float(np.sum(np.where(L & (px > 0.0), ql, 0.0)) - np.sum(np.where(U & ~(px > 0.0), qu, 0.0)))
        pneg = float(np.sum(np.where(L & ~(px > 0.0), ql, 0.0)) - np.sum(np.where(U & (px > 0.0), qu, 0.0)))
        pos, neg, ppos, pneg = pos * beta, neg * beta, ppos * beta, pneg * beta
        for arr, parr in ((v.sw, p.sw), (v.tw, p.tw)):  # :3735-3765
            a, b_ = self._log_split(arr)
            pos += a
            neg += b_
            if arr.size:
                q_ = parr / arr
                ppos += float(np.sum(q_[parr > 0.0]))
                pneg += float(np.sum(q_[~(parr > 0.0)]))
        out = self.comm.allreduce([pos, neg, ppos, pneg])
        pos, neg, ppos, pneg = (float(a) for a in out)
        for i in range(self.c):
            for val, pv in ((v.s[i], p.s[i]), (v.t[i], p.t[i])):
                if val > 1.0:
                    pos += math.log(val)
                else:
                    neg += math.log(val)
                if pv > 0.0:
                    ppos += pv / val
                else:
                    pneg += pv / val
        # evalInfeasDeriv :3465-3509
        dense_infeas = 0.0
        pdense = 0.0
        for i in range(self.c):
            cval = self.cvals[i] - v.s[i] + v.t[i]
            pcval = self.ops.dot(self.Ac[i], p.x) - p.s[i] + p.t[i]
            dense_infeas += cval * cval
            pdense += cval * pcval
        sparse_infeas = 0.0
        psparse = 0.0
        if self.w:  # :3489-3503
            rw1 = self.prob.eval_sparse_con(v.x) - v.sw + v.tw
            sparse_infeas = self.ops.norm(rw1)
            rw2 = np.zeros(self.w)
            self.prob.add_sparse_jacobian(1.0, p.x, rw2)
            rw2 = rw2 - p.sw + p.tw
            psparse = self.ops.dot(rw1, rw2)
        infeas = math.sqrt(dense_infeas + sparse_infeas * sparse_infeas)
        infeas_proj = (pdense + psparse) / infeas if infeas > 0.0 else 0.0
        pTBp = 0.0
        if o["use_diag_hessian"]:  # :3810-3818 (no factor 1/2 here)
            pTBp = float(self.comm.allreduce([float(np.sum(p.x * p.x * self.hdiag))])[0])
        elif self.qn is not None and not o["sequential_linear_method"]:
            xt = self.qn.mult(p.x)
            pTBp = 0.5 * self.ops.dot(xt, p.x)
        merit = (self.fobj + (self.ops.dot(self.gamma_sw, v.sw) + self.ops.dot(self.gamma_tw, v.tw))
                 - self.barrier_param * (pos + neg))
        pmerit = (self.ops.dot(self.g, p.x) + (self.ops.dot(self.gamma_sw, p.sw) + self.ops.dot(self.gamma_tw, p.tw))
                  - self.barrier_param * (ppos + pneg))
        for i in range(self.c):
            merit += self.gamma_s[i] * v.s[i] + self.gamma_t[i] * v.t[i]
            pmerit += self.gamma_s[i] * p.s[i] + self.gamma_t[i] * p.t[i]
        numer = pmerit
        if pTBp > 0.0:
            numer += 0.5 * pTBp
        frac = o["penalty_descent_fraction"]
        small = infeas < 0.1 * o["abs_res_tol"]
        rho_hat = 0.0
        if small:
            denom = -(1.0 - frac) * max_x * infeas
            if numer >= 0.0 and denom < 0.0:
                rho_hat = -(numer / denom)
        else:
            denom = infeas_proj + frac * max_x * infeas
            if numer >= 0.0:
                if denom < 0.0:
                    rho_hat = -(numer / denom)
                else:
                    denom = -(1.0 - frac) * max_x * infeas
                    rho_hat = -(numer / denom)
        if rho_hat > self.rho:
            self.rho = rho_hat
        else:
            self.rho *= 0.5
            if self.rho < rho_hat:
                self.rho = rho_hat
        if self.rho < o["min_rho_penalty_search"]:
            self.rho = o["min_rho_penalty_search"]
        merit += self.rho * infeas
        if small:
            pmerit -= self.rho * max_x * infeas
        else:
            pmerit += self.rho * infeas_proj
        return merit, pmerit

    # ---- line search -----------------------------------------------------------------
    def line_search(self, alpha_min, alpha, m0, dm0):  # :3939-4156
        o = self.opt
        v, p, r = self.vars, self.step, self.res
        fp = o["function_precision"]
        fail = LS_FAILURE
        merit = 0.0
        best_merit = 0.0
        best_alpha = -1.0
        max_it = o["max_line_iters"]
        j = 0
        self.ls_trials = 0
        while j < max_it:
            r.x[:] = self._clamp_step(v.x, alpha, p.x, self.lb, self.ub)
            r.sw[:] = self._clamp_step(v.sw, alpha, p.sw, lower_value=0.0)
            r.tw[:] = self._clamp_step(v.tw, alpha, p.tw, lower_value=0.0)
            r.s[:] = self._clamp_step(v.s, alpha, p.s, lower_value=0.0)
            r.t[:] = self._clamp_step(v.t, alpha, p.t, lower_value=0.0)
            fail_obj, self.fobj, self.cvals = self.prob.eval_obj_con(r.x)
            self.neval += 1
            self.ls_trials += 1
            if fail_obj:
                alpha *= 0.1
                j += 1
                continue
            merit = self.eval_merit_func(self.fobj, self.cvals, r.x, r.s, r.t, r.sw, r.tw)
            if best_alpha < 0.0 or merit < best_merit:
                best_alpha = alpha
                best_merit = merit
            if merit - o["armijo_constant"] * alpha * dm0 < m0 + fp:
                if fail & LS_MIN_STEP:
                    fail = LS_SUCCESS | LS_MIN_STEP
                else:
                    fail = LS_SUCCESS
                if merit <= m0 + fp and merit + fp >= m0:
                    fail |= LS_NO_IMPROVEMENT
                break
            elif fail & LS_MIN_STEP:
                break
            if j < max_it - 1:
                if o["use_backtracking_alpha"]:
                    alpha = 0.5 * alpha
                    if alpha <= alpha_min:
                        alpha = alpha_min
                        fail |= LS_MIN_STEP
                else:
                    alpha_new = -0.5 * dm0 * (alpha * alpha) / (merit - m0 - dm0 * alpha)
                    if alpha_new <= alpha_min:
                        alpha = alpha_min
                        fail |= LS_MIN_STEP
                    elif alpha_new < 0.01 * alpha:
                        alpha = 0.01 * alpha
                    else:
                        alpha = alpha_new
            j += 1
        if j == max_it:
            fail |= LS_MAX_ITERS
        if not (fail & LS_SUCCESS):
            if best_merit <= m0 + fp:
                fail |= LS_SUCCESS
                fail &= ~LS_FAILURE
            elif merit <= m0 + fp and merit + fp >= m0:
                fail |= LS_NO_IMPROVEMENT
            if alpha != best_alpha:
                alpha = best_alpha
                r.x[:] = self._clamp_step(v.x, alpha, p.x, self.lb, self.ub)
                fail_obj, self.fobj, self.cvals = self.prob.eval_obj_con(r.x)
                self.neval += 1
                if fail_obj:
                    fail = LS_FAILURE
            else:
                alpha = best_alpha
        return fail, alpha

    def compute_step_and_update(self, v, alpha, p, eval_obj_con, perform_qn_update):  # :4169-4267
        o = self.opt
        use_qnu = o["use_quasi_newton_update"]
        v.sw[:] = self._clamp_step(v.sw, alpha, p.sw, lower_value=0.0)  # :4177-4183
        v.tw[:] = self._clamp_step(v.tw, alpha, p.tw, lower_value=0.0)
        v.zw[:] = v.zw + alpha * p.zw
        v.zsw[:] = self._clamp_step(v.zsw, alpha, p.zsw, lower_value=0.0)
        v.ztw[:] = self._clamp_step(v.ztw, alpha, p.ztw, lower_value=0.0)
        v.zl[:] = self._clamp_step(v.zl, alpha, p.zl, lower_value=0.0)
        v.zu[:] = self._clamp_step(v.zu, alpha, p.zu, lower_value=0.0)
        v.s[:] = self._clamp_step(v.s, alpha, p.s, lower_value=0.0)
        v.t[:] = self._clamp_step(v.t, alpha, p.t, lower_value=0.0)
        v.z[:] = v.z + alpha * p.z
        v.zs[:] = self._clamp_step(v.zs, alpha, p.zs, lower_value=0.0)
        v.zt[:] = self._clamp_step(v.zt, alpha, p.zt, lower_value=0.0)
        y_qn = None
        if self.qn is not None and perform_qn_update and use_qnu:
            y_qn = -1.0 * self.g
            for i in range(self.c):
                y_qn += v.z[i] * self.Ac[i]
            if self.w:
                self.prob.add_sparse_jacobian_transpose(1.0, v.zw, y_qn)
        v.x[:] = self._clamp_step(v.x, alpha, p.x, self.lb, self.ub)
        if eval_obj_con:
            fail, self.fobj, self.cvals = self.prob.eval_obj_con(v.x)
            self.neval += 1
            if fail:
                return fail
        _, self.g, self.Ac = self.prob.eval_obj_con_gradient(v.x)
        self.ngeval += 1
        update_type = 0
        if self.qn is not None and perform_qn_update and use_qnu:
            s_qn = alpha * p.x
            y_qn += self.g
            for i in range(self.c):
                y_qn += -v.z[i] * self.Ac[i]
            if self.w:
                self.prob.add_sparse_jacobian_transpose(-1.0, v.zw, y_qn)
            update_type = self.qn.update(s_qn, y_qn)
        elif self.qn is not None and perform_qn_update:  # :4261-4263
            update_type = self.qn.update_mult(v.x, v.z, v.zw)
        return update_type

    # ---- starting point --------------------------------------------------------------
    def init_least_squares_multipliers(self, v, r):  # :5366-5534
        o = self.opt
        mb = o["max_bound_value"]
        mu0 = o["init_barrier_param"]
        v.zl[:] = mu0
        v.zu[:] = mu0
        v.z[:] = mu0
        v.s[:] = mu0
        v.t[:] = mu0
        v.zs[:] = mu0
        v.zt[:] = mu0
        for k in ("zw", "sw", "tw", "zsw", "ztw"):
            getattr(v, k)[:] = mu0
        v.zl[self.lb <= -mb] = 0.0
        v.zu[self.ub >= mb] = 0.0
        small = 1e-4
        self.Dinv = np.ones(self.n)
        self._factor(v, np.full(self.w, small))
        c = self.c
        G = np.zeros((c, c))
        for j in range(c):
            xt, _ = self._apply(self.Ac[j])
            for i in range(j, c):
                G[i, j] += self.ops.dot(self.Ac[i], xt)
        for j in range(c):
            for i in range(j + 1, c):
                G[j, i] = G[i, j]
        for i in range(c):
            G[i, i] += small
        self.Glu = sla.lu_factor(G, check_finite=False) if c > 0 else None
        rx = self.g.copy()
        rx += -1.0 * v.zl
        rx += v.zu
        rx *= -1.0
        rzw = np.zeros(self.w) if self.w else None
        yx, _ = self._apply(rx, rzw)
        z = -self.ops.mdot(yx, self.Ac)
        z = self._gsolve(z)
        v.z[:] = z
        for i in range(c):
            rx += z[i] * self.Ac[i]
        _, zw = self._apply(rx, rzw)
        for i in range(c):
            gam = 10.0 * max(self.gamma_s[i], self.gamma_t[i])
            if v.z[i] < -gam or v.z[i] > gam:
                v.z[i] = 0.0
        if self.w:  # :5522-5533
            gam = 10.0 * np.maximum(self.gamma_sw, self.gamma_tw)
            v.zw[:] = np.where((zw < -gam) | (zw > gam), 0.0, zw)

    def init_affine_step_multipliers(self, v, r, p):  # :5536-5656
        o = self.opt
        mb = o["max_bound_value"]
        amin = o["start_affine_multiplier_min"]
        self.init_least_squares_multipliers(v, r)
        v.zl[self.lb <= -mb] = 0.0
        v.zu[self.ub >= mb] = 0.0
        self.compute_kkt_res(v, 0.0, r)
        use_qn = 1
        if o["sequential_linear_method"] or not o["use_qn_gmres_precon"] or o["use_diag_hessian"]:  # :5576
            use_qn = 0
        self.setup_kkt_diag_system(v, use_qn)
        self.setup_kkt_system(v, use_qn)
        self.compute_kkt_step(v, r, p, use_qn)
        v.z[:] = v.z + p.z
        v.s[:] = np.maximum(amin, np.abs(v.s + p.s))
        v.t[:] = np.maximum(amin, np.abs(v.t + p.t))
        v.zs[:] = np.maximum(amin, np.abs(v.zs + p.zs))
        v.zt[:] = np.maximum(amin, np.abs(v.zt + p.zt))
        if self.w:  # :5601-5628
            v.zw[:] = v.zw + p.zw
            v.sw[:] = np.maximum(amin, np.abs(v.sw + p.sw))
            v.tw[:] = np.maximum(amin, np.abs(v.tw + p.tw))
            v.zsw[:] = np.maximum(amin, np.abs(v.zsw + p.zsw))
            v.ztw[:] = np.maximum(amin, np.abs(v.ztw + p.ztw))
        L, U = self._masks()
        v.zl[:] = np.where(L, np.maximum(amin, np.abs(v.zl + p.zl)), v.zl)
        v.zu[:] = np.where(U, np.maximum(amin, np.abs(v.zu + p.zu)), v.zu)
        self.barrier_param = self.compute_comp(v)

    # ---- the major iteration ---------------------------------------------------------
    def _kkt_step_with_refinement(self, v, barrier_for_res, use_qn, inexact_newton_step=0):
        self.compute_kkt_step(v, self.res, self.step, use_qn)
        for _ in range(self.opt["iterative_refinement_steps"]):  # :4985-4991
            self.compute_kkt_res(v, barrier_for_res, self.res)
            self.add_kkt_res_step(v, self.step, self.res, inexact_newton_step)
            self.compute_kkt_step(v, self.res, self.refine, use_qn)
            self.step.add(self.refine)

    def optimize(self):  # :4399-5333 (quasi-Newton branch, monotone / complementarity-fraction)
        o = self.opt
        abs_res_tol = o["abs_res_tol"]
        fprec = o["function_precision"]
        v = self.vars
        barrier_strategy = "monotone"
        input_strategy = o["barrier_strategy"]
        mehrotra_names = ("mehrotra", "mehrotra_predictor_corrector")
        self.barrier_param = o["init_barrier_param"]
        self.rho = o["init_rho_penalty_search"]
        self.niter = self.neval = self.ngeval = self.nhvec = 0
        if self.qn is None and not o["sequential_linear_method"] and not o["use_diag_hessian"]:
            return 1
        self._init_and_check_bounds()
        fail, self.fobj, self.cvals = self.prob.eval_obj_con(v.x)
        self.neval += 1
        if fail:
            return fail
        _, self.g, self.Ac = self.prob.eval_obj_con_gradient(v.x)
        self.ngeval += 1
        sp = o["starting_point_strategy"]
        if sp == "affine_step":
            self.init_affine_step_multipliers(v, self.res, self.step)
        elif sp == "least_squares_multipliers":
            self.init_least_squares_multipliers(v, self.res)
        if self.qn is not None and not o["use_quasi_newton_update"]:  # :4571-4573
            self.qn.update_mult(v.x, v.z, v.zw)
        fobj_prev = 0.0
        alpha_prev = alpha_xprev = alpha_zprev = 0.0
        dm0_prev = 0.0
        no_merit_improvement = 0
        line_search_test = 0
        line_search_failed = 0
        res_norm_prev = 0.0
        info = ""
        self.trace = []
        k = 0
        while k < o["max_major_iters"]:
            qn_reset = 0
            if self.qn is not None and not o["sequential_linear_method"]:
                if k > 0 and k % o["hessian_reset_freq"] == 0 and o["use_quasi_newton_update"]:
                    self.qn.reset()
                    qn_reset = 1
            if self.hook is not None:
                self.hook(self, k)
            rel_function_test = (
                alpha_xprev == 1.0
                and alpha_zprev == 1.0
                and abs(self.fobj - fobj_prev) < o["rel_func_tol"] * abs(fobj_prev)
            )
            if no_merit_improvement:
                line_search_test += 1
            else:
                line_search_test = 0
            comp = self.compute_comp(v)
            monotone_converged = 0
            if barrier_strategy == "monotone":
                self.compute_kkt_res(v, self.barrier_param, self.res)
                mp, md, mi, res_norm = self.compute_res_norm(self.res)
                if k > 0 and (res_norm < 10.0 * self.barrier_param or rel_function_test or line_search_test >= 2):
                    monotone_converged = 1
                if monotone_converged:
                    if self.barrier_param > 0.1 * abs_res_tol:
                        line_search_test = 0
                    mu_frac = o["monotone_barrier_fraction"] * self.barrier_param
                    mu_pow = math.pow(self.barrier_param, o["monotone_barrier_power"])
                    new_mu = min(mu_frac, mu_pow) if mu_pow < mu_frac else mu_frac
                    if new_mu < 0.1 * abs_res_tol:
                        new_mu = 0.09999 * abs_res_tol
                    self.compute_kkt_res(v, new_mu, self.res)
                    mp, md, mi, res_norm = self.compute_res_norm(self.res)
                    self.rho = o["min_rho_penalty_search"]
                    self.barrier_param = new_mu
            elif barrier_strategy in mehrotra_names:  # :4737-4746
                self.compute_kkt_res(v, self.barrier_param, self.res)
                mp, md, mi, res_norm = self.compute_res_norm(self.res)
            else:  # complementarity_fraction :4747-4762
                self.barrier_param = o["monotone_barrier_fraction"] * comp
                if self.barrier_param < 0.1 * abs_res_tol:
                    self.barrier_param = 0.1 * abs_res_tol
                self.compute_kkt_res(v, self.barrier_param, self.res)
                mp, md, mi, res_norm = self.compute_res_norm(self.res)
            self.trace.append(
                dict(
                    iter=k,
                    neval=self.neval,
                    ngeval=self.ngeval,
                    alpha=alpha_prev,
                    alpha_x=alpha_xprev,
                    alpha_z=alpha_zprev,
                    fobj=self.fobj,
                    max_prime=mp,
                    max_infeas=mi,
                    max_dual=md,
                    mu=self.barrier_param,
                    comp=comp,
                    dmerit=dm0_prev,
                    rho=self.rho,
                    info=info,
                )
            )
            converged = 0
            if (
                k > 0
                and self.barrier_param <= 0.1 * abs_res_tol
                and (res_norm < abs_res_tol or rel_function_test or line_search_test >= 2)
            ):
                converged = 1
            if converged:
                break
            gmres_iters = 0
            inexact_newton_step = 0
            if o["use_hvec_product"]:  # :4853-4900
                if res_norm_prev == 0.0:
                    gmres_rtol = float("inf")
                else:
                    gmres_rtol = o["eisenstat_walker_gamma"] * math.pow(res_norm / res_norm_prev,
                                                                         o["eisenstat_walker_alpha"])
                nk = o["nk_switch_tol"]
                if mp < nk and md < nk and mi < nk and gmres_rtol < o["max_gmres_rtol"]:
                    use_qn = 1
                    if o["sequential_linear_method"] or not o["use_qn_gmres_precon"]:
                        use_qn = 0
                    self.setup_kkt_diag_system(v, use_qn)
                    self.setup_kkt_system(v, use_qn)
                    gmres_iters = self.compute_kkt_gmres_step(v, self.res, self.step, gmres_rtol, o["gmres_atol"], use_qn)
                    if gmres_iters < 0:
                        self.compute_kkt_res(v, self.barrier_param, self.res)
                        mp, md, mi, res_norm = self.compute_res_norm(self.res)
                    else:
                        inexact_newton_step = 1
            fobj_prev = self.fobj
            res_norm_prev = res_norm
            seq_linear_step = 0
            diagonal_qn_step = 0
            use_qn = 1
            if inexact_newton_step:
                pass
            elif o["sequential_linear_method"]:
                use_qn = 0
            elif line_search_failed and not o["use_quasi_newton_update"]:  # :4923-4939
                # fixed quasi-Newton approximation and a failed line search: sequential linear step,
                # or only the diagonal b0 of the approximation when it is positive
                use_qn = 0
                seq_linear_step = 1
                if self.qn is not None and self.qn.get_compact()[0] > 0.0:
                    seq_linear_step = 0
                    diagonal_qn_step = 1
            elif o["use_diag_hessian"]:  # :4940-4948
                use_qn = 0
                self.hdiag = self.prob.hessian_diag(v.x, v.z, v.zw if self.w else None)
            mu_for_res = self.barrier_param
            if not inexact_newton_step and barrier_strategy in mehrotra_names:  # affine residual :4958-4964
                mu_for_res = 0.0
                self.compute_kkt_res(v, 0.0, self.res)
                self.compute_res_norm(self.res)
            if not inexact_newton_step:
                self.setup_kkt_diag_system(v, 1 if diagonal_qn_step else use_qn)  # :4968-4980
                self.setup_kkt_system(v, 1 if diagonal_qn_step else use_qn)
                self._kkt_step_with_refinement(v, mu_for_res, use_qn)
            if not inexact_newton_step and barrier_strategy in mehrotra_names:  # :4999-5052
                max_x, max_z = self.compute_max_step(v, 1.0, self.step)
                comp_affine = self.compute_comp_step(v, max_x, max_z, self.step)
                s1 = comp_affine / comp
                sigma = max(s1 * s1 * s1, 0.01)
                self.barrier_param = sigma * comp
                if self.barrier_param < 0.09999 * abs_res_tol:
                    self.barrier_param = 0.09999 * abs_res_tol
                self.compute_kkt_res(v, self.barrier_param, self.res)
                mp, md, mi, res_norm = self.compute_res_norm(self.res)
                if barrier_strategy == "mehrotra_predictor_corrector":
                    self.add_mehrotra_corrector_residual(self.step, self.res)
                    self.compute_kkt_step(v, self.res, self.step, use_qn)
                else:
                    self._kkt_step_with_refinement(v, self.barrier_param, use_qn)
            tau = max(o["min_fraction_to_boundary"], 1.0 - self.barrier_param)
            ceq_step, alpha_x, alpha_z = self.scale_kkt_step(v, self.step, tau, comp, inexact_newton_step)
            alpha = 1.0
            line_fail = LS_FAILURE
            update_type = 0
            line_search_skipped = 0
            no_merit_improvement = 0
            if o["use_line_search"]:
                m0, dm0 = self.eval_merit_init_deriv(v, self.step, alpha_x)
                dm0_prev = dm0
                if 0.0 <= dm0 <= fprec:
                    line_search_skipped = 1
                    update_type = self.compute_step_and_update(v, alpha, self.step, 1, 1)
                    if fobj_prev + fprec <= self.fobj and self.fobj + fprec <= fobj_prev:
                        line_fail = LS_NO_IMPROVEMENT
                else:
                    if dm0 >= 0.0:  # :5130-5173
                        if self.qn is not None:
                            qn_reset = 1
                            self.qn.reset()
                        self.compute_kkt_res(v, self.barrier_param, self.res)
                        mp, md, mi, res_norm = self.compute_res_norm(self.res)
                        diagonal_qn_step = 1
                        self.setup_kkt_diag_system(v, 1)
                        self._kkt_step_with_refinement(v, self.barrier_param, 1, inexact_newton_step)
                        ceq_step, alpha_x, alpha_z = self.scale_kkt_step(v, self.step, tau, comp)
                        m0, dm0 = self.eval_merit_init_deriv(v, self.step, alpha_x)
                        dm0_prev = dm0
                    if dm0 >= 0.0:
                        line_fail = LS_FAILURE
                    else:
                        px_norm = self.ops.maxabs(self.step.x)
                        alpha_min = 1.0
                        if px_norm != 0.0:
                            alpha_min = fprec / px_norm
                        if alpha_min > 0.5:
                            alpha_min = 0.5
                        line_fail, alpha = self.line_search(alpha_min, alpha, m0, dm0)
                        if px_norm < o["design_precision"]:
                            line_fail |= LS_SHORT_STEP
                        if not (line_fail & LS_FAILURE):
                            update_type = self.compute_step_and_update(v, alpha, self.step, 0, 1)
            else:
                m0, dm0 = self.eval_merit_init_deriv(v, self.step, alpha_x)
                dm0_prev = dm0
                line_fail = LS_SUCCESS
                update_type = self.compute_step_and_update(v, alpha, self.step, 1, 1)
                m1 = self.eval_merit_func(self.fobj, self.cvals, v.x, v.s, v.t, v.sw, v.tw)
                if m1 <= m0 + fprec and m1 + fprec >= m0:
                    line_fail |= LS_NO_IMPROVEMENT
                elif abs(dm0) <= fprec:
                    line_fail = LS_NO_IMPROVEMENT
            no_merit_improvement = int(
                bool(line_fail & (LS_NO_IMPROVEMENT | LS_MIN_STEP | LS_SHORT_STEP | LS_FAILURE))
            )
            line_search_failed = line_fail & LS_FAILURE
            alpha_prev, alpha_xprev, alpha_zprev = alpha, alpha_x, alpha_z
            if self.qn is not None and o["use_quasi_newton_update"] and (line_fail & LS_FAILURE):
                qn_reset = 1
                self.qn.reset()
            toks = []
            if gmres_iters != 0:
                toks.append("iNK%d" % gmres_iters)
            if update_type == 1:
                toks.append("dampH")
            elif update_type == 2:
                toks.append("skipH")
            if qn_reset:
                toks.append("resetH")
            if line_fail & LS_FAILURE:
                toks.append("LFail")
            if line_fail & LS_MIN_STEP:
                toks.append("LMnStp")
            if line_fail & LS_MAX_ITERS:
                toks.append("LMxItr")
            if line_fail & LS_NO_IMPROVEMENT:
                toks.append("LNoImprv")
            if seq_linear_step:
                toks.append("SLP")
            if diagonal_qn_step:
                toks.append("DQN")
            if line_search_skipped:
                toks.append("LSkip")
            if ceq_step:
                toks.append("cmpEq")
            info = " ".join(toks)
            if monotone_converged:
                barrier_strategy = input_strategy
            k += 1
            self.niter += 1
        return 0

    # ---- state snapshot in the layout of oracle/ref_driver.cpp ----------------------
    def snapshot(self):
        v = self.vars
        d = dict(
            mu=self.barrier_param,
            rho=self.rho,
            fobj=self.fobj,
            c=self.cvals.copy(),
            z=v.z.copy(),
            s=v.s.copy(),
            t=v.t.copy(),
            zs=v.zs.copy(),
            zt=v.zt.copy(),
            counters=np.array([self.niter, self.neval, self.ngeval]),
            norms=np.array([self.ops.norm(v.x), self.ops.norm(v.zl), self.ops.norm(v.zu)]),
            x=v.x.copy(),
            zl=v.zl.copy(),
            zu=v.zu.copy(),
        )
        if self.w:
            d["wnorms"] = np.array([self.ops.norm(getattr(v, k)) for k in ("zw", "sw", "tw", "zsw", "ztw")])
            for k in ("zw", "sw", "tw", "zsw", "ztw"):
                d[k] = getattr(v, k).copy()
        # integer bookkeeping of SURVEY 8a' in the layout of oracle/ref_driver.cpp: LAPACK (1-based) pivot rows of
        # the last factorizations, and the number of entries sitting exactly at their clamp values
        if getattr(self, "Glu", None) is not None:
            d["gpiv"] = np.asarray(self.Glu[1], dtype=np.int64) + 1
        if self.qn is not None and getattr(self.qn, "lu", None) is not None and len(self.qn.Z) > 0:
            d["mfpiv"] = np.asarray(self.qn.lu[1], dtype=np.int64) + 1
        eps = self.opt["design_precision"]
        loc = np.array([np.sum(v.x == self.lb + eps), np.sum(v.x == self.ub - eps), np.sum(v.zl == eps),
                        np.sum(v.zu == eps)], dtype=float)
        tot = self.comm.allreduce(loc)
        d["clamped"] = np.array([int(tot[0]), int(tot[1]), int(tot[2]), int(tot[3]), int(np.sum(v.s == eps)),
                                 int(np.sum(v.t == eps)), int(np.sum(v.zs == eps)), int(np.sum(v.zt == eps))])
        d["check_flag"] = getattr(self, "check_flag", 0)
        if self.qn is not None:
            b0, d0, M, Z = self.qn.get_compact()
            d["qn_size"] = len(Z)
            d["qn_b0"] = b0
            d["qn_d0"] = np.array(d0).copy()
            d["qn_M"] = np.array(M).copy()
        return d
