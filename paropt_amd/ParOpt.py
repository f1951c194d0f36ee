"""
``from paropt_amd import ParOpt`` -- the calling conventions of the reference's Python layer
(``paropt.ParOpt``, paropt/ParOpt.pyx:761-1521) over the MI355X library, so that a problem
written for the reference runs with the import line changed:

    class Quadratic(ParOpt.Problem):
        def __init__(self, ...):
            super().__init__(MPI.COMM_WORLD, nvars=n, ncon=1)
        def getVarsAndBounds(self, x, lb, ub): x[:] = ...; lb[:] = ...; ub[:] = ...
        def evalObjCon(self, x): return fail, fobj, con
        def evalObjConGradient(self, x, g, A): g[:] = ...; A[0][:] = ...; return fail
    opt = ParOpt.Optimizer(problem, {"algorithm": "ip" | "tr", ...}); opt.optimize()
    x, z, zw, zl, zu = opt.getOptimizedPoint()

Callbacks receive PVec objects with the reference's numpy-style item access on host arrays
(``x[:]``, ``g[:] = ...``); the values move to the GPU when the callback returns.  The MPI
communicator argument is accepted and ignored: ranks are those of the paropt_amd Context (one process
per GPU; ``ParOpt.setContext`` to supply one that is already wired to RCCL or to a host callback).
Not provided: the CSR sparse problem form (``rowp``/``cols``).
"""
import numpy as np

from . import api as _api

_ctx = None
dtype = np.float64


def setContext(ctx):
    """Use an existing paropt_amd.Context (e.g. one initialised for multi-GPU)."""
    global _ctx
    _ctx = ctx


def getContext():
    global _ctx
    if _ctx is None:
        _ctx = _api.Context(0)
    return _ctx


class PVec:
    """paropt.ParOpt.PVec (ParOpt.pyx:914-1188): item access on the host mirror plus the reductions."""

    def __init__(self, vec, fresh=False):
        self._v = vec
        self._a = vec.getArray()
        if not fresh:
            vec.syncToHost()

    def __len__(self):
        return len(self._a)

    def __getitem__(self, k):
        return self._a[k]

    def __setitem__(self, k, values):
        self._a[k] = values

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._a, dtype=dtype)

    def _push(self):
        self._v.syncToDevice()

    def zeroEntries(self):
        self._a[:] = 0.0

    def copyValues(self, other):
        self._a[:] = other[:]

    def norm(self):
        self._push()
        return self._v.norm()

    def l1norm(self):
        self._push()
        return self._v.l1norm()

    def maxabs(self):
        self._push()
        return self._v.maxabs()

    def dot(self, other):
        self._push()
        other._push()
        return self._v.dot(other._v)


def _sizes(kw):
    nvars = kw.get("nvars", 0)
    ncon = kw.get("num_dense_constraints", kw.get("ncon", 0))
    nwcon = kw.get("num_sparse_constraints", kw.get("nwcon", 0))
    nineq = kw.get("num_dense_inequalities", kw.get("ninequality", ncon))
    nwineq = kw.get("num_sparse_inequalities", kw.get("nwinequality", nwcon))
    return int(nvars), int(ncon), int(nwcon), int(nineq), int(nwineq)


class Problem(_api.Problem):
    """paropt.ParOpt.Problem (ParOpt.pyx:787-912): ``Problem(comm, nvars=, ncon=, nwcon=, ...)``."""

    def __init__(self, comm=None, **kwargs):
        nvars, ncon, nwcon, nineq, nwineq = _sizes(kwargs)
        rowp, cols = kwargs.get("rowp"), kwargs.get("cols")
        csr = rowp is not None and cols is not None  # CyParOptSparseProblem (ParOpt.pyx:854-881)
        nwblock = max(1, int(kwargs.get("nwblock", 1) or 1))
        self._nwblock = 1 if csr else nwblock
        self.comm = comm
        self._user = dict(gvb=self.getVarsAndBounds)
        self.getVarsAndBounds = self._gvb
        if csr:
            rp = np.asarray(rowp, dtype=np.intc)
            cl = np.asarray(cols, dtype=np.intc)
            rows = np.repeat(np.arange(nwcon), np.diff(rp))
            seval, sgrad = self.evalSparseObjCon, self.evalSparseObjConGradient
            self._user.update(seval=seval, sgrad=sgrad)
            self.evalSparseObjCon = lambda x, sp: seval(_Host(x), _Host(sp))
            self.evalSparseObjConGradient = lambda x, g, A, data: sgrad(
                _Host(x), _Host(g), [_Host(a) for a in A], data)
            # host views of the same callbacks for checkGradients
            chk = dict(data=np.zeros(len(cl)))

            def _grad(x, g, A):
                return sgrad(x, g, A, chk["data"])

            def _wjac(alpha, x, px, out):
                np.add.at(out._a, rows, alpha * chk["data"] * np.asarray(px[:])[cl])

            def _wjact(alpha, x, pzw, out):
                np.add.at(out._a, cl, alpha * chk["data"] * np.asarray(pzw[:])[rows])

            # no inner-product check: Aw C Aw^T is not diagonal when rows overlap
            self._user.update(eval=lambda x: seval(x, _Host(np.zeros(nwcon))), grad=_grad, wjac=_wjac,
                              wjact=_wjact, winner=None)
        else:
            self._user.update(eval=self.evalObjCon, grad=self.evalObjConGradient)
            if nwcon > 0:
                self._user.update(wcon=self.evalSparseCon, wjac=self.addSparseJacobian,
                                  wjact=self.addSparseJacobianTranspose, winner=self.addSparseInnerProduct)
            # route the host-array protocol of the base class to the PVec protocol of the reference
            self.evalObjCon = self._eval
            self.evalObjConGradient = self._grad
            if nwcon > 0:
                self.evalSparseCon = lambda x, out: self._user["wcon"](_Host(x), _Host(out))
                self.addSparseJacobian = lambda a, x, px, out: self._user["wjac"](a, _Host(x), _Host(px), _Host(out))
                self.addSparseJacobianTranspose = lambda a, x, p, out: self._user["wjact"](a, _Host(x), _Host(p), _Host(out))
                self.addSparseInnerProduct = lambda a, x, c, A: self._user["winner"](a, _Host(x), _Host(c), A)
        super().__init__(getContext(), nvars, ncon, nineq, nwcon=nwcon, nwinequality=nwineq,
                         use_lower=kwargs.get("use_lower", True), use_upper=kwargs.get("use_upper", True),
                         rowp=rowp if csr else None, cols=cols if csr else None, nwblock=1 if csr else nwblock)

    def checkGradients(self, dh=1e-6, x=None, check_hvec_product=False):
        """ParOptProblem::checkGradients (src/ParOptProblem.cpp:376-620) on the host: directional
        finite differences of the objective and the dense constraints against the gradients, and for
        sparse constraints the transpose-equivalence and inner-product identities.  Prints the
        reference's report and returns the relative errors."""
        n, m, w = self.nvars, self.ncon, self.nwcon
        xa, lb, ub = np.zeros(n), np.zeros(n), np.zeros(n)
        self._user["gvb"](_Host(xa), _Host(lb), _Host(ub))
        if x is not None:
            xa = np.array(x[:], dtype=float)
        px = np.array([1.0 if i % 2 == 0 else -1.0 for i in range(n)])
        g, A = np.zeros(n), [np.zeros(n) for _ in range(m)]
        _, f0, c0 = self._user["eval"](_Host(xa.copy()))
        c0 = np.array(c0[:m], dtype=float)
        self._user["grad"](_Host(xa.copy()), _Host(g), [_Host(a) for a in A])
        _, f1, c1 = self._user["eval"](_Host(xa + dh * px))
        c1 = np.array(c1[:m], dtype=float)
        out = {}
        pobj, fd = float(g @ px), (f1 - f0) / dh
        out["objective"] = abs(pobj - fd) / max(abs(fd), 1e-300)
        print("Objective gradient test\nObjective FD: %15.8e  Actual: %15.8e  Err: %8.2e  Rel err: %8.2e" % (
            fd, pobj, abs(fd - pobj), out["objective"]))
        for i in range(m):
            pc, fdc = float(A[i] @ px), (c1[i] - c0[i]) / dh
            out["con%d" % i] = abs(pc - fdc) / max(abs(fdc), 1e-300)
            print("Con[%3d]   FD: %15.8e  Actual: %15.8e  Err: %8.2e  Rel err: %8.2e" % (
                i, fdc, pc, abs(fdc - pc), out["con%d" % i]))
        if w > 0:
            zw = np.array([1.05 + 0.25 * (i % 21) for i in range(w)])
            cw = np.zeros(w)
            self._user["wjac"](1.0, _Host(xa), _Host(px), _Host(cw))
            gt = np.zeros(n)
            self._user["wjact"](1.0, _Host(xa), _Host(zw), _Host(gt))
            d1, d2 = float(zw @ cw), float(gt @ px)
            out["transpose"] = abs(d1 - d2) / max(abs(d2), 1e-300)
            print("\nTranspose-equivalence\nx^{T}*(J(x)*p): %8.2e  p*(J(x)^{T}*x): %8.2e  Err: %8.2e  Rel Err: %8.2e" % (
                d1, d2, abs(d1 - d2), out["transpose"]))
        if w > 0 and self._user.get("winner") is not None:
            cvec = np.array([0.05 + 0.25 * (i % 37) for i in range(n)])
            B = getattr(self, "_nwblock", 1)
            Cw = np.zeros(w * (B + 1) // 2)
            self._user["winner"](1.0, _Host(xa), _Host(cvec), Cw)
            t = np.zeros(n)
            self._user["wjact"](1.0, _Host(xa), _Host(zw), _Host(t))
            t *= cvec
            cw2 = np.zeros(w)
            self._user["wjac"](1.0, _Host(xa), _Host(t), _Host(cw2))
            if B == 1:
                quad = float(np.sum(Cw * zw * zw))
            else:  # packed upper blocks: (i, j), i <= j, of block b at b B (B+1)/2 + i + j (j+1)/2
                quad, incr = 0.0, B * (B + 1) // 2
                for b in range(w // B):
                    for j in range(B):
                        for i in range(j + 1):
                            v = Cw[b * incr + i + j * (j + 1) // 2] * zw[b * B + i] * zw[b * B + j]
                            quad += v if i == j else 2.0 * v
            d1, d2 = float(cw2 @ zw), quad
            out["inner_product"] = abs(d1 - d2) / max(abs(d2), 1e-300)
            print("\nJ(x)*C^{-1}*J(x)^{T} test: \nProduct: %8.2e  Matrix: %8.2e  Err: %8.2e  Rel Err: %8.2e" % (
                d1, d2, abs(d1 - d2), out["inner_product"]))
        return out

    def _gvb(self, x, lb, ub):
        self._user["gvb"](_Host(x), _Host(lb), _Host(ub))

    def _eval(self, x):
        return self._user["eval"](_Host(x))

    def _grad(self, x, g, A):
        return self._user["grad"](_Host(x), _Host(g), [_Host(a) for a in A])


class _Host:
    """A numpy array dressed as a PVec for the callbacks (same item access / reductions)."""

    def __init__(self, a):
        self._a = a

    def __len__(self):
        return len(self._a)

    def __getitem__(self, k):
        return self._a[k]

    def __setitem__(self, k, values):
        self._a[k] = values

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self._a, dtype=dtype)

    def zeroEntries(self):
        self._a[:] = 0.0

    def copyValues(self, other):
        self._a[:] = other[:]

    def norm(self):
        return float(np.sqrt(np.dot(self._a, self._a)))

    def l1norm(self):
        return float(np.sum(np.abs(self._a)))

    def maxabs(self):
        return float(np.max(np.abs(self._a))) if len(self._a) else 0.0

    def dot(self, other):
        return float(np.dot(self._a, other[:]))


def _split_options(options):
    opts = dict(options or {})
    algorithm = opts.pop("algorithm", "tr")
    if opts.get("output_file", "") is None:
        opts["output_file"] = ""
    if opts.get("tr_output_file", "") is None:
        opts["tr_output_file"] = ""
    return algorithm, opts


_TR_ONLY = ("tr_", "filter_")
_MMA_ONLY = ("mma_",)


class InteriorPoint(_api.InteriorPoint):
    """paropt.ParOpt.InteriorPoint (ParOpt.pyx:1229-1365)."""

    def __init__(self, problem, options=None):
        algorithm, opts = _split_options(options)
        opts = {k: v for k, v in opts.items() if not k.startswith(_TR_ONLY + _MMA_ONLY)}
        super().__init__(problem, opts)

    def getOptimizedSlacks(self):
        """s, t, sw, tw as the reference returns them (ParOpt.pyx:1291-1322)."""
        s, t, zs, zt = super().getOptimizedSlacks()
        w = self.getOptimizedSparse()
        return s, t, (PVec(w[1]) if w else None), (PVec(w[2]) if w else None)

    def getOptimizedPoint(self):
        x, z, zl, zu = super().getOptimizedPoint()
        w = self.getOptimizedSparse()
        return (PVec(x), z, (PVec(w[0]) if w else None), PVec(zl) if zl is not None else None,
                PVec(zu) if zu is not None else None)


class TrustRegionSubproblem:
    """paropt.ParOpt.TrustRegionSubproblem (ParOpt.pyx:1395-1413): base of the subproblem wrappers; ``subproblem`` is
    the library object."""

    subproblem = None

    def checkGradients(self, dh=1e-6, x=None, check_hvec_product=False):
        return self.problem.checkGradients(dh, x, check_hvec_product)


class QuadraticSubproblem(TrustRegionSubproblem):
    """paropt.ParOpt.QuadraticSubproblem(problem, qn=None) (ParOpt.pyx:1415-1422)."""

    def __init__(self, problem, qn=None):
        self.problem = problem
        self.subproblem = _api.QuadraticSubproblem(problem, qn)


class TrustRegion:
    """paropt.ParOpt.TrustRegion(subproblem, options) with optimize(InteriorPoint) (ParOpt.pyx:1424-1459)."""

    def __init__(self, prob, options=None):
        algorithm, opts = _split_options(options)
        opts = {k: v for k, v in opts.items() if k != "ip_checkpoint_file" and not k.startswith(_MMA_ONLY)}
        self.prob = prob
        self.tr = _api.TrustRegion(prob.subproblem, opts)

    def optimize(self, optimizer):
        self.tr.optimize(optimizer)

    def getOptimizedPoint(self):
        return PVec(self.prob.subproblem.getLinearModel()[0])


class Optimizer:
    """paropt.ParOpt.Optimizer (ParOpt.pyx:1461-1521, src/ParOptOptimizer.cpp:65-206)."""

    def __init__(self, problem, options=None):
        self.problem = problem
        self.algorithm, self.options = _split_options(options)
        self.ip = self.tr = self.mma = None
        self.subproblem = None
        if self.algorithm not in ("ip", "tr", "mma"):
            raise ValueError("ParOptOptimizer Error: Unrecognized algorithm option %s" % self.algorithm)

    def setTrustRegionSubproblem(self, prob):
        """ParOptOptimizer::setTrustRegionSubproblem (src/ParOptOptimizer.cpp:226-237): algorithm 'tr' then drives the
        caller's subproblem (e.g. ParOptEig.EigenSubproblem) instead of a QuadraticSubproblem built from the options."""
        self.subproblem = prob
        self.ip = self.tr = None

    def optimize(self):
        if self.algorithm == "tr" and self.subproblem is not None:  # :158-183 with the caller's subproblem
            if self.ip is None:
                self.ip = InteriorPoint(self.subproblem.subproblem, self.options)
            if self.tr is None:
                self.tr = TrustRegion(self.subproblem, dict(self.options, algorithm="tr"))
            self.tr.optimize(self.ip)
            return
        if self.algorithm == "ip":
            if self.ip is None:
                self.ip = InteriorPoint(self.problem, self.options)
            ckpt = self.options.get("ip_checkpoint_file")
            self.ip.optimize(ckpt if ckpt else None)
        elif self.algorithm == "tr":
            if self.tr is None:
                opts = {k: v for k, v in self.options.items()
                        if k != "ip_checkpoint_file" and not k.startswith(_MMA_ONLY)}
                self.tr = _api.TrustRegion(self.problem, opts)
            self.tr.optimize()
        else:
            if self.mma is None:
                opts = {k: v for k, v in self.options.items()
                        if k != "ip_checkpoint_file" and not k.startswith(_TR_ONLY)}
                if opts.get("mma_output_file", "") is None:
                    opts["mma_output_file"] = ""
                self.mma = _api.MMA(self.problem, opts)
            self.mma.optimize()

    def getOptimizedPoint(self):
        if self.subproblem is not None and self.tr is not None:  # :209-213
            _, z, zw, zl, zu = self.ip.getOptimizedPoint()
            return self.tr.getOptimizedPoint(), z, zw, zl, zu
        if self.mma is not None:
            x, z, zw, zl, zu = self.mma.getOptimizedPoint()
            return PVec(x), z, (PVec(zw) if zw is not None else None), PVec(zl), PVec(zu)
        if self.tr is not None:
            x, z, zw = self.tr.getOptimizedPoint()
            return PVec(x), z, (PVec(zw) if zw is not None else None), None, None
        return self.ip.getOptimizedPoint()


SKIP_NEGATIVE_CURVATURE, DAMPED_UPDATE = 0, 1


class LBFGS(_api.LBFGS):
    """paropt.ParOpt.LBFGS(prob, subspace=10, update_type=SKIP_NEGATIVE_CURVATURE) (ParOpt.pyx:1210-1219); the
    library's own (ctx, n, subspace) form is accepted too."""

    def __init__(self, prob, *args, subspace=10, update_type=SKIP_NEGATIVE_CURVATURE, **kw):
        if isinstance(prob, _api.Context):
            super().__init__(prob, *args, **kw)
        else:
            if args:
                subspace = args[0]
            super().__init__(prob.ctx, prob.nvars, subspace, update_type)


class LSR1(_api.LSR1):
    """paropt.ParOpt.LSR1(prob, subspace=10) (ParOpt.pyx:1221-1227)."""

    def __init__(self, prob, *args, subspace=10, **kw):
        if isinstance(prob, _api.Context):
            super().__init__(prob, *args, **kw)
        else:
            if args:
                subspace = args[0]
            super().__init__(prob.ctx, prob.nvars, subspace)


def unpack_checkpoint(filename, full=False):
    """Decode a solution file of writeSolutionFile (src/ParOptInteriorPoint.cpp:883-972): returns
    ``barrier, s, z, x, zl, zu`` like paropt.ParOpt.unpack_checkpoint (paropt/ParOpt.pyx:318-354) - whose offsets
    predate the t, zs, zt blocks of the file and therefore misread everything after ``s``; this one follows the
    layout the C++ side writes.  With ``full=True`` a dict with every block (t, zs, zt, zw, sw too)."""
    raw = open(filename, "rb").read()
    nvars, nwcon, ncon = (int(v) for v in np.frombuffer(raw[:12], dtype="<i4"))
    pay = np.frombuffer(raw[12:], dtype="<f8")
    need = 1 + 5 * ncon + 3 * nvars + 2 * nwcon
    if len(pay) < need:
        raise ValueError("solution file %s is too short for its header (%d < %d doubles)" % (filename, len(pay), need))
    out = {"barrier": float(pay[0]), "nvars": nvars, "nwcon": nwcon, "ncon": ncon}
    off = 1
    for key in ("s", "t", "z", "zs", "zt"):
        out[key] = pay[off:off + ncon].copy()
        off += ncon
    for key in ("x", "zl", "zu"):
        out[key] = pay[off:off + nvars].copy()
        off += nvars
    for key in ("zw", "sw"):
        out[key] = pay[off:off + nwcon].copy()
        off += nwcon
    if full:
        return out
    return out["barrier"], out["s"], out["z"], out["x"], out["zl"], out["zu"]


def unpack_output(filename):
    """Columns of an interior-point output file (ParOpt.pyx:61-133): (names, list of arrays)."""
    args = ["iter", "nobj", "ngrd", "nhvc", "alpha", "alphx", "alphz", "fobj", "|opt|", "|infes|", "|dual|", "mu",
            "comp", "dmerit", "rho"]
    cols = [[] for _ in args]
    with open(filename) as fp:
        for line in fp:
            p = line.split()
            if len(p) >= len(args) and p[0].isdigit():
                for i in range(len(args)):
                    try:
                        cols[i].append(int(p[i]) if i < 4 else float(p[i]))
                    except ValueError:
                        cols[i].append(0 if i < 4 else 0.0)
    return args, [np.array(c, dtype=np.int32 if i < 4 else float) for i, c in enumerate(cols)]


def unpack_tr_output(filename):
    """Columns of a trust-region output file (ParOpt.pyx:135-206)."""
    args = ["iter", "fobj", "infeas", "l1", "linfty", "|x - xk|", "tr", "rho", "mod red.", "avg z", "max z",
            "avg pen.", "max pen.", "time(s)"]
    cols = [[] for _ in args]
    with open(filename) as fp:
        for line in fp:
            p = line.split()
            if len(p) >= len(args) and p[0].isdigit():
                for i in range(len(args)):
                    cols[i].append(int(p[i]) if i == 0 else float(p[i]))
    return args, [np.array(c, dtype=np.int32 if i == 0 else float) for i, c in enumerate(cols)]
