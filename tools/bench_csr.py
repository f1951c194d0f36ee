"""
The CSR form of the sparse constraints at scale: convex workload with n variables,
c dense constraints and the rank-local overlapping chain constraints cw_i = 1 - sum_{k<span} x[i*stride+k]^2
(examples/rosenbrock/sparse_rosenbrock.cpp generalised), i.e. a sparse SPD Schur complement with w ~ n/stride
rows that is assembled, factored and solved on the GPU every interior-point iteration.  No CPU reference leg:
the reference's sparse Cholesky needs METIS, which this image lacks.  Prints one JSON line.

    python tools/bench_csr.py [--nglobal 4000000] [--ncon 4] [--span 2] [--stride 1] [--steps 20] [--warmup 5]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nglobal", dest="n", type=int, default=4_000_000)
    ap.add_argument("--ncon", type=int, default=4)
    ap.add_argument("--problem", default="convex")
    ap.add_argument("--span", type=int, default=2)
    ap.add_argument("--stride", type=int, default=1)
    ap.add_argument("--qn-size", type=int, default=10)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    a = ap.parse_args()
    import paropt_amd as pa

    ctx = pa.Context(0)
    t0 = time.perf_counter()
    prob = pa.SeparableProblem(ctx, a.problem, a.n, a.ncon, 0).setChain(a.span, a.stride)
    t_sym = time.perf_counter() - t0
    opts = {"qn_type": "bfgs", "qn_subspace_size": a.qn_size, "abs_res_tol": 1e-30, "abs_step_tol": 0.0,
            "starting_point_strategy": "affine_step", "start_affine_multiplier_min": 0.01, "penalty_gamma": 1000.0,
            "max_major_iters": a.warmup + a.steps, "write_output_frequency": 0}
    ip = pa.InteriorPoint(prob, opts)
    marks = {}

    def cb(k):
        if k in (a.warmup, a.warmup + a.steps):
            ctx.synchronize()
            marks[k] = time.perf_counter()

    ip.setIterationCallback(cb)
    ip.optimize()
    ctx.synchronize()
    niter = ip.getIterationCounters()[0]
    t1 = marks.get(a.warmup + a.steps, time.perf_counter())
    steps = min(niter, a.warmup + a.steps) - a.warmup
    dt = t1 - marks[a.warmup]
    print(json.dumps({
        "metric": "interior-point iterations/s (CSR sparse constraints)", "value": steps / dt,
        "unit": "IP iterations/s", "n_gpus": 1, "steps": steps, "warmup": a.warmup, "ms_per_step": 1e3 * dt / steps,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": "%s n=%d c=%d chain span=%d stride=%d (w=%d) L-BFGS(%d)" % (
            a.problem, a.n, a.ncon, a.span, a.stride, prob.nwcon, a.qn_size)},
        "symbolic_seconds": t_sym, "factor_info": pa.quasidef_factor_info(prob),
    }))


if __name__ == "__main__":
    main()
