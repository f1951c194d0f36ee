"""
TEST INFRASTRUCTURE ONLY -- numpy model of the *fused* KKT algebra used by the HIP product.

The product (paropt_amd/csrc/ip.cpp) does not mirror the reference's operation sequence: it
assembles both Schur complements from ONE weighted Gram matrix of the panel P = [Ac | Z] and
turns every bordered solve into "one panel-dot pass + tiny host algebra + one panel-axpy
pass" (DESIGN.md "Fused KKT step").  This file states that algebra in numpy, on top of the
mirror-mode oracle, so that tests can check on the CPU that it is the same map as the
reference's computeKKTStep / iterative refinement (src/ParOptInteriorPoint.cpp:2634-2737,
4985-4991) before any GPU is involved.  It is never imported by the product.
"""
import numpy as np
import scipy.linalg as sla

from . import paropt_oracle as po


class FusedInteriorPoint(po.InteriorPoint):
    # --- "kernels" --------------------------------------------------------------------
    def _bound_terms(self, v):
        L, U = self._masks()
        xl = np.where(L, v.x - self.lb, 1.0)
        xu = np.where(U, self.ub - v.x, 1.0)
        return L, U, xl, xu

    def _panel(self, use_qn):
        Z = []
        if self.qn is not None and use_qn:
            Z = list(self.qn.get_compact()[3])
        return list(self.Ac) + Z, len(Z)

    def setup_kkt_diag_system(self, v, use_qn):
        """Dinv + ONE weighted Gram W = P^T diag(Dinv) P (replaces :1932-1950 and :2648-2654)."""
        o = self.opt
        L, U, xl, xu = self._bound_terms(v)
        b0 = self.qn.b0 if (self.qn is not None and use_qn) else 0.0
        d = np.full(self.n, b0 + o["qn_sigma"])
        d = d + np.where(L, v.zl / xl, 0.0)
        d = d + np.where(U, v.zu / xu, 0.0)
        self.Dinv = 1.0 / d
        P, k = self._panel(use_qn)
        c = self.c
        m = len(P)
        W = np.zeros((m, m))
        for i in range(m):
            t = self.Dinv * P[i]
            W[i, :] = self.comm.allreduce(np.array([np.dot(t, P[j]) for j in range(m)])) if m else 0.0
        self.W = W
        G = W[:c, :c].copy()
        for i in range(c):
            G[i, i] += v.s[i] / v.zs[i] + v.t[i] / v.zt[i]
        self.Glu = sla.lu_factor(G, check_finite=False) if c > 0 else None
        self._wk = k

    def setup_kkt_system(self, v, use_qn):
        """Ce = W_ZZ - W_ZA G^-1 W_AZ - M / (d0 d0^T)  (SURVEY.md 3.4 identity check)."""
        self.Celu = None
        if self.qn is None or not use_qn:
            return
        b0, d0, M, Z = self.qn.get_compact()
        k = len(Z)
        if k == 0:
            return
        c = self.c
        assert self.W.shape[0] == c + k, "W must have been assembled with the same panel"
        Wzz = self.W[c:, c:]
        Waz = self.W[:c, c:]
        Ce = Wzz - (Waz.T @ self._gsolve(Waz) if c > 0 else 0.0) - M / np.outer(d0, d0)
        self.Celu = sla.lu_factor(Ce, check_finite=False)

    def _fused_solve(self, v, d1, dense_rhs, use_qn):
        """K0^-1 then SMW correction with one dot pass and one axpy pass over the panel.

        d1        : n-vector right-hand side after elimination of the bound rows
        dense_rhs : c-vector  b.z + (b.zs + s b.s)/zs - (b.zt + t b.t)/zt   (zero for bx-only)
        returns px and the coefficient pair (yz_total, zeta)
        """
        P, k = self._panel(use_qn)
        c = self.c
        t = self.Dinv * d1
        dots = self.ops.mdot(t, P) if P else np.zeros(0)
        a_t, z_t = dots[:c], dots[c:]
        yz = self._gsolve(dense_rhs - a_t)
        zeta = np.zeros(k)
        yz2 = np.zeros(c)
        if k > 0:
            Waz = self.W[:c, c:]
            Wzz = self.W[c:, c:]
            ztp = z_t + Waz.T @ yz  # Z^T px0 without a second pass
            zeta = sla.lu_solve(self.Celu, ztp, check_finite=False)
            yz2 = self._gsolve(-(Waz @ zeta))
        alpha = np.concatenate([yz - yz2, -zeta])
        acc = d1.copy()
        for j in range(len(P)):
            acc += alpha[j] * P[j]
        return self.Dinv * acc, yz, yz2

    def _dense_blocks(self, v, b, yz, yz2, p):
        """Host algebra of solveKKTDiagSystem (full, :2150-2170) minus the bx-only solve (:2289-2305)."""
        zs1 = yz - b.s
        zt1 = -b.t - yz
        p.z[:] = yz - yz2
        p.zs[:] = zs1 - yz2
        p.zt[:] = zt1 + yz2
        p.s[:] = (b.zs - v.s * zs1) / v.zs + (v.s * yz2) / v.zs
        p.t[:] = (b.zt - v.t * zt1) / v.zt + (v.t * (-yz2)) / v.zt

    def compute_kkt_step(self, v, r, p, use_qn):
        L, U, xl, xu = self._bound_terms(v)
        d1 = r.x + np.where(L, r.zl / xl, 0.0) - np.where(U, r.zu / xu, 0.0)
        rhs = r.z + (r.zs + v.s * r.s) / v.zs - (r.zt + v.t * r.t) / v.zt
        px, yz, yz2 = self._fused_solve(v, d1, rhs, use_qn)
        p.x[:] = px
        self._dense_blocks(v, r, yz, yz2, p)
        p.zl[:] = np.where(L, (r.zl - v.zl * px) / xl, 0.0)
        p.zu[:] = np.where(U, (r.zu + v.zu * px) / xu, 0.0)
