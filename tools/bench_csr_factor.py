"""
Factor / solve timing of the device sparse Cholesky on a 2-D pattern (one sparse constraint per edge of an
nx x ny grid of variables: S is the Laplacian-like matrix of the line graph, separators are dense cliques).
Run once as is and once with PAROPT_AMD_NO_FRONTS=1 to see what the dense fronts buy.

    python tools/bench_csr_factor.py [--nx 300] [--ny 300] [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nx", type=int, default=300)
    ap.add_argument("--ny", type=int, default=300)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    import paropt_amd as pa
    from csr_helpers import grid_pattern

    n = a.nx * a.ny
    rowp, cols = grid_pattern(a.nx, a.ny)
    w = len(rowp) - 1
    rng = np.random.default_rng(0)
    data = rng.uniform(0.5, 1.5, size=int(rowp[-1]))
    ctx = pa.Context(0)

    class P(pa.Problem):
        def __init__(self):
            super().__init__(ctx, n, 1, 1, nwcon=w, nwinequality=w, rowp=rowp, cols=cols)

        def getVarsAndBounds(self, x, lb, ub):
            x[:], lb[:], ub[:] = 0.5, 0.0, 1.0

        def evalSparseObjCon(self, x, sparse):
            sparse[:] = 1.0
            return 0, float(np.sum(x * x)), np.array([1.0 - np.sum(x) / n])

        def evalSparseObjConGradient(self, x, g, A, d):
            g[:] = 2.0 * x
            A[0][:] = -1.0 / n
            d[:] = data
            return 0

    t0 = time.perf_counter()
    prob = P()
    t_sym = time.perf_counter() - t0
    pa.InteriorPoint(prob, {"max_major_iters": 0}).optimize()  # uploads the Jacobian entries

    def vec(arr):
        v = pa.PVec(ctx, len(arr))
        v.from_numpy(arr)
        return v

    x = vec(np.full(n, 0.5))
    d, c = vec(rng.uniform(0.5, 2.0, n)), vec(rng.uniform(0.1, 1.0, w))
    bx, bw = vec(rng.standard_normal(n)), vec(rng.standard_normal(w))
    yx, yw = pa.PVec(ctx, n), pa.PVec(ctx, w)
    pa.quasidef_factor(prob, x, d, c)
    pa.quasidef_apply(prob, x, d, c, bx, bw, yx, yw)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.reps):
        pa.quasidef_factor(prob, x, d, c)
    ctx.synchronize()
    t_fac = (time.perf_counter() - t0) / a.reps
    t0 = time.perf_counter()
    for _ in range(a.reps):
        pa.quasidef_apply(prob, x, d, c, bx, bw, yx, yw)
    ctx.synchronize()
    t_app = (time.perf_counter() - t0) / a.reps
    sym = pa.CsrSymbolic(n, rowp, cols)
    print(json.dumps({"pattern": "grid %dx%d" % (a.nx, a.ny), "n": n, "w": w, "nnzL": sym.nnzL, "levels": sym.nlevels,
                      "fronts": sym.nfronts, "max_front": sym.max_front, "setup_s": t_sym,
                      "factor_ms": 1e3 * t_fac, "apply_ms": 1e3 * t_app,
                      "no_fronts": bool(int(os.environ.get("PAROPT_AMD_NO_FRONTS", "0")))}))


if __name__ == "__main__":
    main()
