#!/usr/bin/env python3
"""
bench.py -- IP iterations/sec of the MI355X-native interior point on BASELINE.json's metric
configuration (config 3: separable random_convex, n = 50 M design variables, m = 32 dense
constraints + bounds, L-SR1(10) Hessian), plus the HBM roofline of the headline kernel
(ParOptVec::mdot over the 32-column dense-constraint panel), a second roofline entry for the
weighted-Gram kernel, same-run stream ceilings and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R]

N = 1 runs in this process.  N > 1 without WORLD_SIZE in the environment: this script starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py` as a CHILD process
(before anything here has touched the GPU), relays rank 0's JSON line and exits with the child's
code.  Under torchrun (WORLD_SIZE set) it is one rank of the job: one process per GPU, the design
vector sharded over the ranks (total n fixed -> strong scaling), reductions over RCCL/xGMI.

A "step" is one major interior-point iteration (KKT residual, Schur-complement assembly, KKT step
+ iterative refinement, step scaling, merit derivative, line search, quasi-Newton update).  One
repeat = one optimize() call of W + K iterations from the same starting point; the K timed
iterations are bracketed by barrier + stream sync on both sides, max over ranks; `value` is the
median over R repeats (min / max beside it).
"""
import argparse
import json
import os
import socket
import statistics
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_GLOBAL = 50_000_000
NCON = 32
QN_SIZE = 10
HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_MFMA_PEAK_TFLOPS = 78.6  # MI355X fp64 matrix peak (spec); the measured sustained rate is in profiles/


def cpu_baseline(n, ncon, iters, log, nwcon=0, nw=0, qn="sr1", qn_size=QN_SIZE, problem="convex"):
    """Reference (oracle/_ref/ref_driver, the unmodified C++ reference + MKL under MPICH) timed on
    this box's host cores on a bounded sample; falls back to the numpy restatement."""
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    mpiexec = "/opt/conda/bin/mpiexec"
    ncpu = os.cpu_count() or 1
    if os.path.exists(drv) and os.path.exists(mpiexec):
        ranks = max(1, min(64, ncpu // 2 if ncpu >= 4 else ncpu))
        env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1",
                   PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
        cmd = [mpiexec, "-n", str(ranks), drv, "bench", "problem=%s" % problem, "n=%d" % n, "c=%d" % ncon,
               "opt.qn_type=%s" % qn, "opt.qn_subspace_size=%d" % qn_size, "opt.abs_res_tol=1e-30",
               "opt.start_affine_multiplier_min=0.01", "opt.max_major_iters=%d" % iters,
               "opt.write_output_frequency=0"]
        if nwcon > 0:
            # the driver's weighting groups are rank-local (as in examples/rosenbrock): pick a rank count
            # whose shards hold whole groups so that the sharded problem IS the global one
            while ranks > 1 and (n % ranks or (n // ranks) % nw or nwcon % ranks):
                ranks -= 1
            cmd[2] = str(ranks)
            cmd += ["nwcon=%d" % (nwcon // ranks), "nw=%d" % nw, "nwstart=0", "nwskip=0"]
        try:
            t0 = time.time()
            out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd="/tmp")
            for ln in out.stdout.splitlines():
                if ln.startswith("{"):
                    r = json.loads(ln)
                    return {"value": r["niter"] / r["seconds"], "unit": "IP iterations/s", "cores": ranks,
                            "kind": "reference", "steady_state": False,
                            "sample": "unmodified reference (MPICH ranks x MKL seq), same problem at n=%d, "
                                      "first %d iterations (quasi-Newton memory ramping 0->%d: cheaper than "
                                      "steady-state iterations, so the GPU/CPU ratio is conservative), optimize() "
                                      "only; wall incl. launch %.1fs" % (n, r["niter"], min(r["niter"], qn_size),
                                                                         time.time() - t0)}
            log("cpu_baseline: reference produced no result: %s" % (out.stderr[-400:],))
        except Exception as e:  # pragma: no cover
            log("cpu_baseline: reference failed: %r" % (e,))
    # numpy restatement ("port"), single core, smaller sample
    from oracle import paropt_oracle as po

    ns = 1_000_000
    opts = {"qn_type": qn, "qn_subspace_size": qn_size, "abs_res_tol": 1e-30,
            "start_affine_multiplier_min": 0.01, "max_major_iters": 6}
    wargs = dict(nwcon=int(nwcon * ns // n), nw=nw, nwstart=0, nwskip=0) if nwcon > 0 else {}
    ip = po.InteriorPoint(po.SepProblem(problem, ns, ncon, **wargs), opts)
    t0 = time.time()
    ip.optimize()
    dt = time.time() - t0
    return {"value": ip.niter / dt * (ns / float(n)), "unit": "IP iterations/s", "cores": 1, "kind": "port",
            "steady_state": False,
            "sample": "numpy oracle, n=%d, %d iterations, rate scaled linearly to n=%d" % (ns, ip.niter, n)}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=12)
    ap.add_argument("--repeats", type=int, default=5, help="optimize() runs; the line reports the median")
    ap.add_argument("--nglobal", dest="n", type=int, default=N_GLOBAL,
                    help="global design variables (default: the metric's 50M)")
    ap.add_argument("--ncon", type=int, default=NCON)
    ap.add_argument("--qn", type=str, default="sr1")
    ap.add_argument("--qn-size", type=int, default=QN_SIZE)
    ap.add_argument("--problem", type=str, default="convex", choices=["convex", "quadratic"])
    ap.add_argument("--nwcon", type=int, default=0,
                    help="config 4: number of sparse weighting constraints (one per group of --nw variables)")
    ap.add_argument("--nw", type=int, default=20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-n", type=int, default=0, help="n of the CPU sample (default: --nglobal)")
    ap.add_argument("--cpu-iters", type=int, default=6)
    ap.add_argument("--no-constant-jacobian", action="store_true",
                    help="only measure the variant whose gradient callback rewrites the (constant) constraint "
                         "Jacobian at every call, as the reference's example problems do")
    ap.add_argument("--skip-copy-variant", action="store_true",
                    help="do not also measure the Jacobian-rewriting variant beside the headline")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N>1: continue on a torch.distributed callback if the native RCCL communicator fails "
                         "(default: exit non-zero)")
    return ap.parse_args(argv)


def launch_children(a, argv):
    """--gpus N > 1 outside torchrun: run the N-rank job as a child process and relay its line."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(a.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    print("[bench] launching: " + " ".join(cmd), file=sys.stderr, flush=True)
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in proc.stdout:
        if ln.startswith("{") and '"metric"' in ln:
            line = ln.strip()
        else:
            sys.stderr.write(ln)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    elif rc == 0:
        rc = 1
    return rc


def stub_rank(a, rank, world):
    """PAROPT_BENCH_STUB=1 (CPU test of the launcher plumbing): no GPU work, gloo only."""
    import torch
    import torch.distributed as dist

    if world > 1:
        dist.init_process_group(backend="gloo")
        t = torch.ones(1, dtype=torch.float64)
        dist.all_reduce(t)
        seen = int(t.item())
    else:
        seen = 1
    if rank == 0:
        print(json.dumps({"metric": "stub", "n_gpus": seen, "steps": a.steps, "warmup": a.warmup, "stub": True,
                          "gpus_arg": a.gpus}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    argv = sys.argv[1:]
    a = parse_args(argv)
    in_job = "WORLD_SIZE" in os.environ
    if a.gpus > 1 and not in_job:
        sys.exit(launch_children(a, argv))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("PAROPT_BENCH_STUB", "0") == "1":
        sys.exit(stub_rank(a, rank, world))

    def log(msg):
        if rank == 0:
            print("[bench] " + msg, file=sys.stderr, flush=True)

    if in_job and a.gpus != world:
        log("warning: --gpus %d but WORLD_SIZE=%d; the job runs on %d ranks" % (a.gpus, world, world))

    import torch

    # test hook for the 1-GPU development box: all ranks share GPU 0 and reduce over gloo
    share_gpu = os.environ.get("PAROPT_BENCH_SHARE_GPU", "0") == "1"
    if share_gpu:
        local_rank = 0
    elif world > 1 and torch.cuda.device_count() < world:
        log("error: %d ranks but only %d visible GPU(s)" % (world, torch.cuda.device_count()))
        sys.exit(2)
    dist = None
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(local_rank)
        if share_gpu:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    import paropt_amd as pa

    ctx = pa.Context(local_rank)
    comm_kind = "self"
    if world > 1 and share_gpu:
        ctx.init_callback_from_torch()
        comm_kind = "gloo callback (shared-GPU test mode)"
    elif world > 1:
        try:
            ctx.init_rccl_from_torch()  # native ncclAllReduce / ncclAllGather on the solver's own stream
            comm_kind = "rccl"
        except Exception as e:
            if not a.allow_fallback:
                log("native RCCL communicator failed (%r); refusing to measure a fallback (--allow-fallback)" % (e,))
                sys.exit(3)
            log("native RCCL communicator failed (%r); falling back to torch.distributed all_gather" % (e,))
            ctx.init_callback_from_torch(device=torch.device("cuda", local_rank))
            comm_kind = "torch.distributed(nccl) callback"
    ranks_seen = ctx.rank_size()[1]  # what the communicator of the solver actually spans
    if ranks_seen != world:
        log("error: the solver's communicator spans %d ranks, the job has %d" % (ranks_seen, world))
        sys.exit(4)

    def barrier_sync():
        ctx.synchronize()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            ctx.synchronize()

    K = a.steps
    # the driver passes --warmup 5: the timed window must not start while the quasi-Newton memory is still
    # filling (iterations get more expensive until it is full), so the warmup is at least qn_size + 2
    W = max(a.warmup, a.qn_size + 2)
    prob = pa.SeparableProblem(ctx, a.problem, a.n, a.ncon, 0)
    if a.nwcon > 0:
        prob.setWeighting(a.nwcon, a.nw, 0, 0)
    opts = {"qn_type": a.qn, "qn_subspace_size": a.qn_size, "abs_res_tol": 1e-30,
            "start_affine_multiplier_min": 0.01, "max_major_iters": W + K, "write_output_frequency": 0}
    ip = pa.InteriorPoint(prob, opts)
    stamp = {}

    def cb(k):
        if k == W:
            barrier_sync()
            stamp["counters0"] = ctx.counters()
            ctx.time_mdot(a.ncon)  # HIP events around every mdot<ncon> launch of the timed region
            ctx.time_wgram(True)
            stamp["t0"] = time.perf_counter()

    ip.setIterationCallback(cb)

    def one_run():
        """One optimize() of W + K iterations; returns (seconds of the K timed iterations, details)."""
        ip.resetQuasiNewtonHessian()
        barrier_sync()
        ip.optimize()
        barrier_sync()
        t1 = time.perf_counter()
        elapsed = t1 - stamp["t0"]
        red1, lau1 = ctx.counters()
        mdot_ms, mdot_n = ctx.time_mdot_result()
        wg = [ctx.time_wgram_result(w) for w in (0, 1)]
        ctx.time_mdot(0)
        ctx.time_wgram(False)
        if dist is not None:
            t = torch.tensor([elapsed], dtype=torch.float64, device="cpu" if share_gpu else "cuda")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        niter, neval, ngeval = ip.getIterationCounters()
        assert niter == W + K, (niter, W, K)
        return elapsed, dict(red=(red1 - stamp["counters0"][0]) / float(K),
                             launches=(lau1 - stamp["counters0"][1]) / float(K), mdot_ms=mdot_ms, mdot_n=mdot_n,
                             wgram=wg, niter=niter, neval=neval, phases=ip.getPhaseTimes())

    def measure(constant_jacobian):
        prob.setLinearConstraints(constant_jacobian)
        runs = [one_run() for _ in range(max(1, a.repeats))]
        times = sorted(r[0] for r in runs)
        med = statistics.median(times)
        det = min(runs, key=lambda r: abs(r[0] - med))[1]
        return med, times, det, runs

    variants = {}
    if not a.no_constant_jacobian:
        variants["constant_jacobian"] = measure(True)
    if a.no_constant_jacobian or not a.skip_copy_variant:
        variants["jacobian_rewritten_every_gradient_call"] = measure(False)
    head = "constant_jacobian" if "constant_jacobian" in variants else "jacobian_rewritten_every_gradient_call"
    elapsed, times, det, runs = variants[head]

    # ---- roofline of the headline kernel, measured live with HIP events on the context stream ----
    nl = prob.nvars
    x = pa.PVec(ctx, nl).fill_hash(1, 10, prob.offset, 2.0, -1.0)
    # the dense-constraint panel the solver itself streams: fresh hash vectors of the same shape
    V = [pa.PVec(ctx, nl).fill_hash(1, 20 + j, prob.offset, 2.0, -1.0) for j in range(a.ncon)]
    ms_isolated, _ = pa.bench_mdot(x, V, 20)
    # same-run stream ceilings: what a read-only stream and a copy reach on this box right now
    ms_ro = pa.bench_stream(x, V[0], 0, 20)
    ms_cp = pa.bench_stream(x, V[0], 1, 20)
    stream = {"read_only_GBps": 16.0 * nl / (ms_ro * 1e-3) * 1e-9, "copy_GBps": 16.0 * nl / (ms_cp * 1e-3) * 1e-9,
              "what": "x.y over two %d-element vectors (read-only) and y <- x (copy), 20 launches each, "
                      "HIP events on the solver's stream, same process as the timed run" % nl}
    # the figure of record: the mdot<ncon> launches the solver itself issued in the timed regions (the constraint
    # evaluations of the line search), HIP events on the launch stream; the isolated loop is kept beside it
    mdot_ms_run = sum(r[1]["mdot_ms"] for r in runs)
    mdot_launches_run = sum(r[1]["mdot_n"] for r in runs)
    ms = mdot_ms_run / mdot_launches_run if mdot_launches_run > 0 else ms_isolated
    alg_bytes = 8.0 * (a.ncon + 1) * nl
    achieved = alg_bytes / (ms * 1e-3) * 1e-9
    traffic, traffic_src = None, None
    for fn in ("r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
        try:  # PMC bytes per launch from committed rocprofv3 passes (NOT measured in this run), this exact shape
            pm = json.load(open(os.path.join(ROOT, "profiles", fn)))
            if pm["n"] == nl and a.ncon == 32:
                k = pm["raw"]["void po::mdot_kernel<32>"]
                traffic = k["hbm_read_bytes_corrected"] + k["hbm_write_bytes"]
                traffic_src = "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes; not collected in this run)" % fn
                break
        except Exception:
            pass
    roofline = {"bound": "hbm", "kernel": "mdot_kernel<32> (ParOptVec::mdot, nvecs=%d, n_local=%d)" % (a.ncon, nl),
                "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
                "traffic": traffic, "traffic_source": traffic_src, "avg_launch_ms": ms,
                "launches_timed_in_run": mdot_launches_run,
                "avg_launch_ms_isolated_loop": ms_isolated, "algorithmic_bytes": alg_bytes,
                "stream_ceiling": stream}
    # second entry: the weighted Gram (the only MFMA kernel), HBM and MFMA fractions of the in-run launches
    second = None
    for w in (1, 0):
        wms = sum(r[1]["wgram"][w][0] for r in runs)
        wn = sum(r[1]["wgram"][w][1] for r in runs)
        if wn > 0:
            cols = runs[0][1]["wgram"][w][2]
            wbytes = sum(r[1]["wgram"][w][3] for r in runs) / wn
            avg = wms / wn
            flops = float(cols) * (cols + 1) * nl  # useful flops of the symmetric product
            second = {"kernel": "wgram_kernel (W = P^T diag(Dinv) P, %d columns%s)" % (
                          cols, ", L-SR1 columns formed in the pass" if w == 1 else ""),
                      "avg_launch_ms": avg, "launches_timed_in_run": wn, "algorithmic_bytes": wbytes,
                      "hbm": {"achieved": wbytes / (avg * 1e-3) * 1e-9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                              "frac": wbytes / (avg * 1e-3) * 1e-9 / HBM_PEAK_GBPS},
                      "mfma": {"achieved": flops / (avg * 1e-3) * 1e-12, "peak": FP64_MFMA_PEAK_TFLOPS,
                               "unit": "TFLOP/s", "frac": flops / (avg * 1e-3) * 1e-12 / FP64_MFMA_PEAK_TFLOPS,
                               "flops": "useful: cols*(cols+1)*n"}}
            break
    roofline["second"] = second

    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline and world == 1:
            # bounded sample (about half a minute of host time): a quarter of the rows of a large workload, the rate
            # scaled back linearly in n -- every pass of the reference is O(n) and memory-bound at these sizes
            cpu_n = a.cpu_n or (a.n // 4 if a.n >= 20_000_000 and a.nwcon == 0 else a.n)
            cpu = cpu_baseline(cpu_n, a.ncon, a.cpu_iters, log, a.nwcon * cpu_n // a.n, a.nw, a.qn, a.qn_size,
                               a.problem)
            if cpu and cpu_n != a.n and cpu.get("kind") == "reference":
                cpu["value"] *= cpu_n / float(a.n)
                cpu["sample"] += "; measured at n=%d, rate scaled by %d/%d to the workload's n (O(n) passes)" % (
                    cpu_n, cpu_n, a.n)
        niter = det["niter"]
        kind, nred, ngat = ctx.comm_info()
        res = {
            "metric": "IP iterations/sec (KKT solve+line search), n=%s vars m=%d, 1/2/4/8 GPUs" % (
                ("%dM" % (a.n // 1000000)) if a.n % 1000000 == 0 else str(a.n), a.ncon),
            "value": K / elapsed,
            "unit": "IP iterations/s",
            "n_gpus": ranks_seen,
            "steps": K,
            "warmup": W,
            "warmup_requested": a.warmup,
            "repeats": len(times),
            "ms_per_step": 1e3 * elapsed / K,
            "ms_per_step_min": 1e3 * times[0] / K,
            "ms_per_step_max": 1e3 * times[-1] / K,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": "%s: separable random_%s n=%d, m=%d dense + bounds, L-%s(%d), "
                                   "design vector sharded over %d GPU(s)" % (
                                       ("config 4 (+%d weighting constraints on groups of %d)" % (a.nwcon, a.nw))
                                       if a.nwcon > 0 else ("config 3" if a.problem == "convex" else "config 2"),
                                       a.problem, a.n, a.ncon, a.qn.upper(), a.qn_size, world),
                       "n_global": a.n, "ncon": a.ncon, "nwcon": a.nwcon, "qn": a.qn,
                       "evals_per_iter": (det["neval"] - 1) / float(niter),
                       "collective": comm_kind, "reductions_per_iter": det["red"],
                       "launches_per_iter": det["launches"],
                       "rccl_allreduce_calls": nred, "rccl_allgather_calls": ngat,
                       "constraint_jacobian": head},
            "variants": {k: {"value": K / v[0], "ms_per_step": 1e3 * v[0] / K,
                             "user_eval_ms_per_iter": 1e3 * v[2]["phases"].get("user_eval", 0.0) / v[2]["niter"]}
                         for k, v in variants.items()},
            "roofline": roofline,
            "cpu_baseline": cpu,
            "phase_ms_per_iter": {k: 1e3 * v / niter for k, v in det["phases"].items()},
            "user_eval_ms_per_iter": 1e3 * det["phases"].get("user_eval", 0.0) / niter,
        }
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
