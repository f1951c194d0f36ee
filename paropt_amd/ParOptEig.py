"""
``from paropt_amd import ParOptEig`` -- the reference's ``paropt.ParOptEig`` (paropt/ParOptEig.pyx:40-121) over the
MI355X library: the compact eigenvalue model of one constraint under the trust-region driver, assembled by the user's
code exactly as the reference's example does (examples/eigenvalue/eigenvalue_opt.py:298-308):

    qn = ParOpt.LBFGS(problem, subspace=10)
    approx = ParOptEig.CompactEigenApprox(problem, N)
    eig_qn = ParOptEig.EigenQuasiNewton(qn, approx, index=0)
    subproblem = ParOptEig.EigenSubproblem(problem, eig_qn)
    subproblem.setUpdateEigenModel(problem.updateModel)      # updateModel(x, approx)
    opt = ParOpt.Optimizer(problem, options); opt.setTrustRegionSubproblem(subproblem); opt.optimize()
"""
import numpy as np

from . import api as _api
from .ParOpt import PVec, TrustRegionSubproblem


class CompactEigenApprox:
    """ParOptEig.CompactEigenApprox(problem, N): c(s) = c0 + g0^T s + 1/2 s^T H M H^T s (ParOptEig.pyx:40-89)."""

    def __init__(self, problem=None, N=None, _wrap=None):
        self.ptr = _wrap if _wrap is not None else _api.CompactEigenApprox(problem, int(N))
        self._vecs = None

    def getApproximationVectors(self):
        """(g0, [h_0 .. h_{N-1}]) as PVec objects with item access on host arrays; what is written through them is
        uploaded when the model-update callback returns."""
        if self._vecs is None:
            self._vecs = (PVec(self.ptr.g0), [PVec(h) for h in self.ptr.hvecs])
        return self._vecs

    def setApproximationValues(self, c=None, M=None, Minv=None):
        if c is not None:
            self.ptr.c0 = float(c)
        if M is not None:
            self.ptr.M[:, :] = np.asarray(M, dtype=float)
        if Minv is not None:
            self.ptr.Minv[:, :] = np.asarray(Minv, dtype=float)

    def _refresh(self):
        """bring the host views up to date with what the library preset (g0) or holds (hvecs)"""
        if self._vecs is not None:
            g0, hs = self._vecs
            g0._v.syncToHost()
            for h in hs:
                h._v.syncToHost()

    def _flush(self):
        """upload what the callback wrote through the vectors' host views"""
        if self._vecs is not None:
            g0, hs = self._vecs
            g0._push()
            for h in hs:
                h._push()


class EigenQuasiNewton(_api.EigenQuasiNewton):
    """ParOptEig.EigenQuasiNewton(qn, eigh, index=0) (ParOptEig.pyx:91-100); qn may be None."""

    def __init__(self, qn, eigh, index=0):
        self.approx = eigh
        super().__init__(qn, eigh.ptr, index)


class EigenSubproblem(TrustRegionSubproblem):
    """ParOptEig.EigenSubproblem(problem, eig) with setUpdateEigenModel(callback) (ParOptEig.pyx:115-130)."""

    def __init__(self, problem, eig):
        self.problem = problem
        self.eig = eig
        self.callback = None
        self.subproblem = _api.EigenSubproblem(problem, eig)

    def setUpdateEigenModel(self, callback):
        """callback(x, approx): x a PVec of the point, approx the CompactEigenApprox handed to EigenQuasiNewton."""
        self.callback = callback
        approx = self.eig.approx

        def _update(x, _lib_approx):
            approx._refresh()
            callback(PVec(x), approx)
            approx._flush()

        self.subproblem.setEigenModelUpdate(_update)
