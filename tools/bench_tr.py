"""
Config 5 of BASELINE.json: the compact-eigenvalue subproblem under the trust-region driver
(n = 5M, N = 10 curvature directions, trust-region defaults) on one MI355X, next to the unmodified
reference (oracle/_ref/ref_driver trbench) on the host cores.  Prints one JSON line in the shape of bench.py's:
`value` = trust-region iterations/s, with the inner interior-point iterations/s, launches and host syncs per inner
iteration, the iteration-level roofline (algorithmic bytes of every n-sized launch / time / 8 TB/s), the HBM
roofline of the model-evaluation mdot (ParOptVec::mdot over [gk | Ak | Z | H | g0 | s]) timed with HIP events on
the solver's stream inside a second, shorter run, and the CPU reference.

    python tools/bench_tr.py [--nglobal 5000000] [--ncon 4] [--eig-N 10] [--tr-iters 10] [--assembly objects]
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0


def cpu_reference(a):
    from bench import host_cpu_budget

    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    mpiexec = "/opt/conda/bin/mpiexec"
    if not (os.path.exists(drv) and os.path.exists(mpiexec)):
        return None
    cpus = host_cpu_budget()
    ranks = max(1, min(64, cpus["usable"]))
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1", PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
    # a bounded sample: a fifth of the rows (every pass of the reference is O(n) and memory-bound: the rate is scaled
    # to the workload's n), the first `cpu_tr_iters` trust-region iterations
    n_s = max(100_000, a.n // 5)
    cmd = [mpiexec, "-n", str(ranks), drv, "trbench", "problem=%s" % a.problem, "n=%d" % n_s, "c=%d" % a.ncon,
           "eig_N=%d" % a.eig_N, "eig_index=0", "eig_curv=%g" % a.curv, "opt.qn_subspace_size=%d" % a.qn_size,
           "opt.max_major_iters=%d" % a.max_major_iters, "tr.tr_max_iterations=%d" % a.cpu_tr_iters]
    t0 = time.time()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1800, cwd="/tmp")
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            r = json.loads(ln)
            return {"value": r["tr_iters"] / r["seconds"] * n_s / float(a.n), "unit": "TR iterations/s", "cores": ranks,
                    "kind": "reference", "sample_n": n_s, "sample_tr_iterations": r["tr_iters"],
                    "sample_seconds": r["seconds"], "host": cpus,
                    "sample": "unmodified reference (%d MPICH ranks x sequential MKL), same problem and model at n=%d, "
                              "first %d trust-region iterations, rate scaled by %d/%d to the workload's n; wall incl. "
                              "launch %.1fs" % (ranks, n_s, r["tr_iters"], n_s, a.n, time.time() - t0)}
    sys.stderr.write(out.stderr[-400:])
    return None


def build(pa, ctx, a, tr_iters):
    prob = pa.SeparableProblem(ctx, a.problem, a.n, a.ncon, 0)
    opts = {"qn_subspace_size": a.qn_size, "tr_max_iterations": tr_iters, "max_major_iters": a.max_major_iters}
    if a.assembly == "objects":
        # the reference's own assembly (examples/eigenvalue/eigenvalue_opt.py:298-308) through the object-level API
        qn = pa.LBFGS(ctx, prob.nvars, a.qn_size)
        approx = pa.CompactEigenApprox(prob, a.eig_N)
        sub = pa.EigenSubproblem(prob, pa.EigenQuasiNewton(qn, approx, 0))
        filled = []

        def upd(x, e):  # the synthetic model of oracle/ref_driver.cpp eig_update, filled on the device
            if not filled:
                for i in range(a.eig_N):
                    e.hvecs[i].fill_hash(0, 300 + i, prob.offset, 2.0, -1.0)
                    e.hvecs[i].scale(1.0 / e.hvecs[i].norm())
                filled.append(1)
            for i in range(a.eig_N):
                for j in range(a.eig_N):
                    d = -a.curv * (1.0 + 0.1 * i)
                    e.M[i, j] = d if i == j else 0.0
                    e.Minv[i, j] = 1.0 / d if i == j else 0.0

        sub.setEigenModelUpdate(upd)
        ip = pa.InteriorPoint(sub, {k: v for k, v in opts.items() if not k.startswith("tr_")})
        tr = pa.TrustRegion(sub, opts)
        return prob, tr, (lambda: tr.optimize(ip)), (sub, qn, approx, ip)
    tr = pa.TrustRegion(prob, opts)
    tr.setEigenModelSynthetic(a.eig_N, 0, 0, a.curv)
    return prob, tr, tr.optimize, None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nglobal", dest="n", type=int, default=5_000_000)
    ap.add_argument("--ncon", type=int, default=4)
    ap.add_argument("--problem", default="quadratic")
    ap.add_argument("--eig-N", dest="eig_N", type=int, default=10)
    ap.add_argument("--curv", type=float, default=2.0)
    ap.add_argument("--qn-size", type=int, default=10)
    ap.add_argument("--tr-iters", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=3, help="whole runs; the line reports the median")
    ap.add_argument("--assembly", default="driver", choices=["driver", "objects"],
                    help="driver: TrustRegion(problem) assembles everything (ParOptOptimizer's way); objects: the "
                         "reference user code's own assembly (LBFGS, CompactEigenApprox, EigenQuasiNewton, "
                         "EigenSubproblem, InteriorPoint(subproblem), TrustRegion(subproblem).optimize(ip))")
    # interior-point iteration cap per subproblem solve, as the reference's trust-region examples set it
    # (examples/topology_optimization/topo_optimization.py:560: 100); the degenerate steering LP can
    # otherwise sit on a failed line search until the default cap of 5000 (same in the reference)
    ap.add_argument("--max-major-iters", type=int, default=200)
    ap.add_argument("--cpu-tr-iters", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    a = ap.parse_args()
    import paropt_amd as pa

    ctx = pa.Context(0)

    def one_run(tr_iters, time_mdot_nv=0):
        prob, tr, run, keep = build(pa, ctx, a, tr_iters)
        counts = []

        def cb(i):
            if i > 0:
                s = tr.getState()
                counts.append((s["subproblem_iters"], s["adaptive_subproblem_iters"]))

        tr.setIterationCallback(cb)
        ctx.synchronize()
        red0, lau0 = ctx.counters()
        by0 = ctx.algorithmic_bytes()[0]
        if time_mdot_nv:
            ctx.time_mdot(time_mdot_nv)
        t0 = time.perf_counter()
        run()
        ctx.synchronize()
        dt = time.perf_counter() - t0
        red1, lau1 = ctx.counters()
        by1 = ctx.algorithmic_bytes()[0]
        md = ctx.time_mdot_result() if time_mdot_nv else (0.0, 0)
        ctx.time_mdot(0)
        s = tr.getState()
        counts.append((s["subproblem_iters"], s["adaptive_subproblem_iters"]))
        ip_iters = sum(x + y for x, y in counts)
        return dict(dt=dt, tr_iters=s["iter_count"], ip_iters=ip_iters, launches=lau1 - lau0, syncs=red1 - red0,
                    bytes=by1 - by0, mdot_ms=md[0], mdot_n=md[1], nvars=prob.nvars)

    one_run(2)  # warm-up: code objects, allocations
    runs = sorted((one_run(a.tr_iters) for _ in range(max(1, a.repeats))), key=lambda r: r["dt"])
    r = runs[len(runs) // 2]
    inner = max(r["ip_iters"], 1)
    # roofline of the headline kernel class: the most frequent ParOptVec::mdot of the run, the steering LP's model
    # evaluation x = the step against [gk | Ak (m)] (TrustRegionSubproblem::evalLinearModel), HIP events on the
    # solver's stream
    nv = 1 + a.ncon
    t = one_run(min(a.tr_iters, 6), time_mdot_nv=nv)
    roofline = None
    if t["mdot_n"] > 0:
        ms = t["mdot_ms"] / t["mdot_n"]
        alg = 8.0 * (nv + 1) * r["nvars"]
        roofline = {"bound": "hbm", "kernel": "mdot_kernel (ParOptVec::mdot of the steering LP's model evaluation, nvecs=%d, n=%d)" % (
                        nv, r["nvars"]),
                    "achieved": alg / (ms * 1e-3) * 1e-9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                    "frac": alg / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS, "traffic": None, "avg_launch_ms": ms,
                    "launches_timed_in_run": t["mdot_n"], "algorithmic_bytes": alg,
                    "note": "timed in a second run of %d trust-region iterations (a timed launch flushes the batch it "
                            "rides in, so the run of record is not instrumented)" % min(a.tr_iters, 6)}
    res = {"metric": "trust-region iterations/s (compact eigenvalue subproblem, SL1QP + adaptive penalty)",
           "value": r["tr_iters"] / r["dt"], "unit": "TR iterations/s", "n_gpus": 1, "tr_iterations": r["tr_iters"],
           "repeats": len(runs), "seconds": r["dt"], "seconds_min": runs[0]["dt"], "seconds_max": runs[-1]["dt"],
           "higher_is_better": True, "vs_baseline": None,
           "inner_ip_iterations": r["ip_iters"], "inner_ip_iterations_per_s": r["ip_iters"] / r["dt"],
           "ms_per_inner_iteration": 1e3 * r["dt"] / inner,
           "launches_per_inner_iteration": r["launches"] / float(inner),
           "host_syncs_per_inner_iteration": r["syncs"] / float(inner),
           # iteration-level roofline: algorithmic bytes of every n-sized launch of the run (problem evaluations, model
           # updates and the trust-region driver's own passes included) per inner iteration
           "iteration_bytes": r["bytes"] / float(inner),
           "iteration_frac": r["bytes"] / r["dt"] * 1e-9 / HBM_PEAK_GBPS,
           "roofline": roofline,
           "dtype": "f64", "data": "synthetic",
           "config": {"workload": "config 5: ParOptEigenSubproblem under ParOptTrustRegion, separable random_%s "
                                  "n=%d, m=%d, N=%d curvature directions, L-BFGS(%d), trust-region defaults, "
                                  "max_major_iters=%d per subproblem solve" % (
                                      a.problem, a.n, a.ncon, a.eig_N, a.qn_size, a.max_major_iters),
                      "assembly": a.assembly},
           "cpu_baseline": None if a.no_cpu_baseline else cpu_reference(a)}
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
