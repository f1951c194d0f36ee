"""GPU: ownership.  The C ABI keeps the reference's intrusive reference counts (src/ParOptVec.h:28-47): after the
last decref / destroy of everything a scenario created, the number of live device vectors and the HBM bytes behind
them are back where they started (po_live_objects) - for vectors, quasi-Newton objects, every problem form
(built-in, weighting, CSR, callback), the interior point with all its branches, the trust-region driver with the
eigenvalue model, and MMA."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _scenarios(ctx):
    import paropt_amd as pa

    def vectors():
        v = [pa.PVec(ctx, 1000) for _ in range(5)]
        v[0].fill_hash(0, 1, 0, 1.0, 0.0)
        v[0].mdot(v[1:])
        v[0].getArray()  # pinned host mirror

    def quasi_newton():
        for cls in (pa.LBFGS, pa.LSR1):
            qn = cls(ctx, 500, 4)
            s, y = pa.PVec(ctx, 500).fill_hash(0, 1, 0, 1.0, 0.0), pa.PVec(ctx, 500).fill_hash(0, 2, 0, 1.0, 0.1)
            for _ in range(6):
                qn.update(s, y)
            qn.mult(s, y)
            qn.getCompactMat()

    def interior_point():
        for kind, qn in (("quadratic", "bfgs"), ("convex", "sr1"), ("rosenbrock", "bfgs")):
            ip = pa.InteriorPoint(pa.SeparableProblem(ctx, kind, 2000, 3), {"qn_type": qn, "max_major_iters": 8})
            ip.optimize()
            ip.getOptimizedPoint()

    def sparse_forms():
        ip = pa.InteriorPoint(pa.SeparableProblem(ctx, "convex", 2000, 2).setWeighting(100, 5), {"max_major_iters": 8})
        ip.optimize()
        ip.getOptimizedSparse()
        ip = pa.InteriorPoint(pa.SeparableProblem(ctx, "convex", 2000, 2).setChain(3, 2), {"max_major_iters": 8})
        ip.optimize()
        ip = pa.InteriorPoint(pa.SeparableProblem(ctx, "convex", 600, 2).setChain(2, 1),
                              {"max_major_iters": 12, "use_hvec_product": True, "gmres_subspace_size": 5,
                               "nk_switch_tol": 1e3, "max_gmres_rtol": 1.0})
        ip.optimize()

    def callbacks():
        class Q(pa.Problem):
            def __init__(self):
                super().__init__(ctx, 50, 1, 1)

            def getVarsAndBounds(self, x, lb, ub):
                x[:], lb[:], ub[:] = 0.3, -1.0, 1.0

            def evalObjCon(self, x):
                return 0, float(np.sum(x * x)), np.array([np.sum(x) - 1.0])

            def evalObjConGradient(self, x, g, A):
                g[:] = 2.0 * x
                A[0][:] = 1.0
                return 0

        ip = pa.InteriorPoint(Q(), {"max_major_iters": 10})
        ip.optimize()

    def trust_region():
        tr = pa.TrustRegion(pa.SeparableProblem(ctx, "quadratic", 800, 2), {"tr_max_iterations": 4, "qn_subspace_size": 4})
        tr.setEigenModelSynthetic(3, 0, 0, 1.5)
        tr.optimize()
        tr.getOptimizedPoint()
        tr2 = pa.TrustRegion(pa.SeparableProblem(ctx, "convex", 400, 2).setChain(2, 1),
                             {"tr_max_iterations": 3, "tr_accept_step_strategy": "filter_method"})
        tr2.optimize()

    def mma():
        m = pa.MMA(pa.SeparableProblem(ctx, "convex", 500, 2), {"mma_max_iterations": 3})
        m.optimize()
        m.getAsymptotes()

    def sparse_callbacks():
        # host-side sparse callbacks (block form, nwblock 1 and 2): the panel columns handed to addSparseJacobian
        # borrow solver memory and get a pinned mirror each time -- none may outlive the call
        for nwblock in (1, 2):
            class W(pa.Problem):
                def __init__(self):
                    super().__init__(ctx, 40, 1, 1, nwcon=8, nwinequality=8, nwblock=nwblock)

                def getVarsAndBounds(self, x, lb, ub):
                    x[:], lb[:], ub[:] = 0.1, 0.0, 1.0

                def evalObjCon(self, x):
                    return 0, float(np.sum((x - 0.4) ** 2)), np.array([np.sum(x) - 2.0])

                def evalObjConGradient(self, x, g, A):
                    g[:] = 2.0 * (x - 0.4)
                    A[0][:] = 1.0
                    return 0

                def evalSparseCon(self, x, out):
                    out[:] = 1.0 - x.reshape(8, 5).sum(axis=1)

                def addSparseJacobian(self, alpha, x, px, out):
                    out[:] -= alpha * px.reshape(8, 5).sum(axis=1)

                def addSparseJacobianTranspose(self, alpha, x, pzw, out):
                    out[:] -= alpha * np.repeat(pzw, 5)

                def addSparseInnerProduct(self, alpha, x, cvec, A):
                    d = cvec.reshape(8, 5).sum(axis=1)
                    if nwblock == 1:
                        A[:] += alpha * d
                    else:  # packed upper 2 x 2 blocks: constraints 2b and 2b+1 do not share variables
                        A[0::3] += alpha * d[0::2]
                        A[2::3] += alpha * d[1::2]

            ip = pa.InteriorPoint(W(), {"max_major_iters": 6, "qn_subspace_size": 3})
            ip.optimize()

    return [vectors, quasi_newton, interior_point, sparse_forms, callbacks, sparse_callbacks, trust_region, mma]


def test_every_object_gives_its_memory_back():
    import paropt_amd as pa

    ctx = pa.Context(0)
    gc.collect()
    base = pa.live_objects()
    base_mirrors = pa.live_host_mirrors()
    for scenario in _scenarios(ctx):
        scenario()
        gc.collect()
        ctx.synchronize()
        assert pa.live_objects() == base, (scenario.__name__, pa.live_objects(), base)
        assert pa.live_host_mirrors() == base_mirrors, (scenario.__name__, pa.live_host_mirrors(), base_mirrors)
    # and while objects are alive the counters do move
    v = pa.PVec(ctx, 1000)
    assert pa.live_objects()[0] == base[0] + 1 and pa.live_objects()[1] > base[1]
    del v
    gc.collect()
    assert pa.live_objects() == base
    ctx.close()
