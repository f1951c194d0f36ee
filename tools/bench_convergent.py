#!/usr/bin/env python3
"""The CONVERGENT comparison BASELINE.md section 4 prescribes for config 3: "C3 is a fixed-iteration throughput
measurement; the convergent comparison uses L-BFGS on the same data".

Config 3's data (separable convex, c = 32 dense constraints + bounds) with L-BFGS, run to `abs_res_tol` on the GPU at
the workload's n: major iterations, function / gradient evaluations, seconds of optimize(), final objective and
residual norms.  Beside it the unmodified reference (oracle/_ref/ref_driver) on the host cores at a SAMPLE n (the same
problem family at a size its run fits the budget), same options: iterations and seconds to the same tolerance.
One JSON line.

    python tools/bench_convergent.py [--n 50000000] [--ncon 32] [--qn-size 20] [--tol 1e-6] [--cpu-n 5000000]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def gpu_run(n, ncon, qn_size, tol, max_iters, repeats):
    import paropt_amd as pa

    ctx = pa.Context(0)
    opts = {"qn_type": "bfgs", "qn_subspace_size": qn_size, "abs_res_tol": tol, "start_affine_multiplier_min": 0.01,
            "max_major_iters": max_iters, "write_output_frequency": 0}
    prob = pa.SeparableProblem(ctx, "convex", n, ncon, 0)
    ip = pa.InteriorPoint(prob, opts)
    runs = []
    for _ in range(repeats):
        ip.resetQuasiNewtonHessian()
        ctx.synchronize()
        red0, lau0 = ctx.counters()
        by0 = ctx.algorithmic_bytes()[0]
        t0 = time.perf_counter()
        ip.optimize()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        red1, lau1 = ctx.counters()
        niter, neval, ngeval = ip.getIterationCounters()
        hist = ip.getHistory()
        last = [ln for ln in hist.splitlines() if ln[:5].strip().isdigit()][-1].split()
        msgs = [ln.strip() for ln in hist.splitlines() if ln.startswith("ParOpt")]
        runs.append({"seconds": dt, "niter": niter, "neval": neval, "ngeval": ngeval, "fobj": ip.getObjective()[0],
                     "converged": "Successfully converged" in hist,
                     "termination": msgs[-1] if msgs else ("max_major_iters reached" if niter >= max_iters else ""),
                     "final_opt_infeas_dual": [float(last[8]), float(last[9]), float(last[10])],
                     "mu": ip.getBarrierParameter(), "host_syncs": red1 - red0, "launches": lau1 - lau0,
                     "algorithmic_GB": (ctx.algorithmic_bytes()[0] - by0) * 1e-9})
    runs.sort(key=lambda r: r["seconds"])
    med = runs[len(runs) // 2]
    med["all_seconds"] = [r["seconds"] for r in runs]
    med["it_per_s"] = med["niter"] / med["seconds"]
    med["achieved_TBps"] = med["algorithmic_GB"] / med["seconds"] * 1e-3
    return med


def cpu_run(n, ncon, qn_size, tol, max_iters, budget_s):
    sys.path.insert(0, ROOT)
    from bench import host_cpu_budget

    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(drv) and os.path.exists(mpiexec)):
        return {"error": "oracle/_ref/ref_driver is not present on this box"}
    cpus = host_cpu_budget()
    ranks = max(1, min(64, cpus["usable"]))
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
    cmd = [mpiexec, "-n", str(ranks), drv, "bench", "problem=convex", "n=%d" % n, "c=%d" % ncon, "opt.qn_type=bfgs",
           "opt.qn_subspace_size=%d" % qn_size, "opt.abs_res_tol=%g" % tol, "opt.start_affine_multiplier_min=0.01",
           "opt.max_major_iters=%d" % max_iters]
    t0 = time.time()
    try:
        out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=budget_s, cwd="/tmp")
    except subprocess.TimeoutExpired:
        return {"error": "the reference did not finish within %.0f s at n = %d" % (budget_s, n), "ranks": ranks}
    res = {"n": n, "ranks": ranks, "wall_s": time.time() - t0, "host": cpus}
    for ln in out.stdout.splitlines():
        if ln.startswith("{") and "niter" in ln:
            r = json.loads(ln)
            res.update(niter=r["niter"], neval=r.get("neval"), ngeval=r.get("ngeval"), seconds=r["seconds"],
                       fobj=r.get("fobj"), it_per_s=r["niter"] / r["seconds"])
    if "niter" not in res:
        res["error"] = out.stderr[-400:]
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=50_000_000)
    ap.add_argument("--ncon", type=int, default=32)
    ap.add_argument("--qn-size", type=int, default=20)
    ap.add_argument("--tol", type=float, default=1e-6)
    ap.add_argument("--max-iters", type=int, default=6000)
    ap.add_argument("--repeats", type=int, default=3)
    ap.add_argument("--cpu-n", type=int, default=5_000_000)
    ap.add_argument("--cpu-budget", type=float, default=900.0)
    ap.add_argument("--no-cpu", action="store_true")
    a = ap.parse_args()
    gpu = gpu_run(a.n, a.ncon, a.qn_size, a.tol, a.max_iters, a.repeats)
    # the same problem at the CPU sample size on the GPU too: iteration counts depend (mildly) on n
    gpu_small = gpu_run(a.cpu_n, a.ncon, a.qn_size, a.tol, a.max_iters, 1) if not a.no_cpu else None
    cpu = None if a.no_cpu else cpu_run(a.cpu_n, a.ncon, a.qn_size, a.tol, a.max_iters, a.cpu_budget)
    out = {"what": "config 3's data (separable convex, c = %d + bounds) with L-BFGS(%d) run to abs_res_tol = %g "
                   "(BASELINE.md section 4: the convergent comparison)" % (a.ncon, a.qn_size, a.tol),
           "gpu": dict(gpu, n=a.n), "gpu_at_cpu_sample_n": (dict(gpu_small, n=a.cpu_n) if gpu_small else None),
           "cpu_reference": cpu}
    if cpu and "seconds" in cpu and gpu_small:
        out["time_to_tolerance_ratio_at_sample_n"] = cpu["seconds"] / gpu_small["seconds"]
        out["cpu_seconds_scaled_to_n"] = cpu["seconds"] * a.n / float(a.cpu_n)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
