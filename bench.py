#!/usr/bin/env python3
"""
bench.py -- IP iterations/sec of the MI355X-native interior point on BASELINE.json's metric
configuration (config 3: separable random_convex, n = 50 M design variables, m = 32 dense
constraints + bounds, L-SR1(10) Hessian), the HBM roofline of the headline kernel
(ParOptVec::mdot over the 32-column dense-constraint panel) and of the whole iteration, a second
roofline entry for the weighted-Gram kernel, same-run stream ceilings and a CPU baseline.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--boundary builtin|facade|both]

`value` is measured under the REFERENCE's problem contract (src/ParOptProblem.h:146-189): the gradient callback
rewrites the whole constraint Jacobian at every call.  The variant with the library's opt-in declaration that the
dense constraints are linear (po_problem_set_linear_constraints, an EXTENSION of the reference API) is reported
beside it under `variants`, never as `value`.  --boundary facade makes `value` the same workload implemented as a
USER's ParOptProblem subclass outside the library (examples/random_convex_amd.cpp on include/ParOptAMD.hpp); the
default run reports that measurement under `boundary`.

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
import gc
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
USER_LIB = os.path.join(ROOT, "examples", "librandom_convex_user.so")
WEIGHTING_LIB = os.path.join(ROOT, "examples", "libweighting_user.so")  # config 4 as a user's ParOptSparseProblem


# ------------------------------------------------------------------------------------------------
# CPU baseline
# ------------------------------------------------------------------------------------------------
def host_cpu_budget():
    """Cores this process may really use: scheduler affinity, capped by the cgroup CPU quota (v2 cpu.max or v1
    cfs quota) -- os.cpu_count() ignores both, and oversubscribed busy-polling MPI ranks are 5x slower."""
    info = {"os_cpu_count": os.cpu_count() or 1}
    try:
        info["affinity"] = len(os.sched_getaffinity(0))
    except Exception:
        info["affinity"] = info["os_cpu_count"]
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            a, b = f.read().split()
            if a != "max":
                quota = float(a) / float(b)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
                q = float(f.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
                p = float(f.read())
            if q > 0:
                quota = q / p
        except Exception:
            pass
    info["cgroup_quota_cpus"] = quota
    usable = info["affinity"]
    if quota is not None:
        usable = max(1, min(usable, int(quota)))
    info["usable"] = usable
    try:
        info["loadavg_1min"] = os.getloadavg()[0]
    except Exception:
        info["loadavg_1min"] = None
    return info


def reference_traffic_bytes(n, c, k):
    """Bytes one iteration of the REFERENCE's operation sequence moves at quasi-Newton width k (SURVEY.md 3.4/8d,
    validated there against the compiled reference): 8 n (332 + 48 c + c^2 + 45 k + 5 c k + 2 k^2)."""
    return 8.0 * n * (332 + 48 * c + c * c + 45 * k + 5 * c * k + 2 * k * k)


def run_reference(drv, mpiexec, ranks, n, ncon, iters, qn, qn_size, problem, nwcon, nw, timeout):
    """One optimize() of the unmodified reference; returns (niter, seconds of optimize(), seconds of every major
    iteration, wall seconds incl. launch, error text)."""
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1",
               PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
    cmd = [mpiexec, "-n", str(ranks), drv, "bench", "problem=%s" % problem, "n=%d" % n, "c=%d" % ncon,
           "opt.qn_type=%s" % qn, "opt.qn_subspace_size=%d" % qn_size, "opt.abs_res_tol=1e-30",
           "opt.start_affine_multiplier_min=0.01", "opt.max_major_iters=%d" % iters]
    if nwcon > 0:
        cmd += ["nwcon=%d" % (nwcon // ranks), "nw=%d" % nw, "nwstart=0", "nwskip=0"]
    t0 = time.time()
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd="/tmp")
    wall = time.time() - t0
    niter, secs, per_iter = 0, 0.0, []
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            r = json.loads(ln)
            if "niter" in r:
                niter, secs = r["niter"], r["seconds"]
            if "iteration_seconds" in r:
                per_iter = r["iteration_seconds"]
    return niter, secs, per_iter, wall, (None if niter > 0 else out.stderr[-400:])


def run_reference_mdot(drv, mpiexec, ranks, n, nvecs, reps, timeout):
    """ParOptBasicVec::mdot of the unmodified reference alone (src/ParOptVec.cpp:152-170): seconds per call."""
    env = dict(os.environ, MKL_NUM_THREADS="1", OMP_NUM_THREADS="1",
               PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
    cmd = [mpiexec, "-n", str(ranks), drv, "mdot", "n=%d" % n, "nvecs=%d" % nvecs, "reps=%d" % reps]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd="/tmp")
    for ln in out.stdout.splitlines():
        if ln.startswith("{"):
            return json.loads(ln)["seconds"]
    return None


def cpu_baseline(n, ncon, iters, log, nwcon=0, nw=0, qn="sr1", qn_size=QN_SIZE, problem="convex", budget_s=60.0,
                 full_size=False):
    """The unmodified reference (oracle/_ref/ref_driver: the reference's C++ + MKL under MPICH) timed on this box's
    host cores to the protocol of BASELINE.md section 4: K = qn_size + 8 major iterations from a cold quasi-Newton
    memory (abs_res_tol = 1e-30), optimize() only; WHOLE-RUN rate and STEADY-STATE rate (the iterations that run with
    full memory, from the driver's per-iteration stamps); best of up to three runs at one size (the hosts are
    shared); plus ParOptVec::mdot alone at nvecs 8, 32, 40.  The sample is bounded: a probe at n/25 sizes the largest
    n <= workload n / 4 whose runs fit the budget; every pass of the reference is O(n) and memory-bound at these
    sizes, so rates are scaled linearly to the workload's n (said in `sample`).  Falls back to the numpy restatement."""
    drv = os.path.join(ROOT, "oracle", "_ref", "ref_driver")
    mpiexec = "/opt/conda/bin/mpiexec"
    cpus = host_cpu_budget()
    K = max(iters, qn_size + 8)
    kfull = qn_size + 1  # iterations k >= kfull run with full memory (k pairs are held at iteration k)
    if os.path.exists(drv) and os.path.exists(mpiexec):
        ranks = max(1, min(64, cpus["usable"]))
        if nwcon > 0:
            # the driver's weighting groups are rank-local (as in examples/rosenbrock): pick a rank count
            # whose shards hold whole groups so that the sharded problem IS the global one
            while ranks > 1 and (n % ranks or (n // ranks) % nw or nwcon % ranks):
                ranks -= 1
        try:
            # untimed tiny run first: pages in the driver, MKL and MPICH on a fresh box
            run_reference(drv, mpiexec, ranks, 20_000 * ranks, ncon, 2, qn, qn_size, problem, 0, nw, 300)
            t_start = time.time()

            def steady(per_iter):
                tail = per_iter[kfull:]
                return (len(tail) / sum(tail)) if len(tail) >= 4 and sum(tail) > 0 else None

            runs = []
            n_probe = max(200_000, n // 25) if nwcon == 0 else max(nw * ranks * 1000, n // 10)
            if nwcon > 0:
                n_probe -= n_probe % (nw * ranks)
            wfrac = (nwcon / float(n)) if nwcon else 0.0

            def one(n_s):
                niter, secs, per_iter, wall, err = run_reference(
                    drv, mpiexec, ranks, n_s, ncon, K, qn, qn_size, problem,
                    int(round(wfrac * n_s)) if nwcon else 0, nw, 900)
                if niter <= 0:
                    log("cpu_baseline: reference produced no result: %s" % (err,))
                    raise RuntimeError("no result")
                runs.append({"n": n_s, "iterations": niter, "seconds": secs, "wall_s": wall,
                             "whole_run_it_per_s": niter / secs, "steady_state_it_per_s": steady(per_iter),
                             "iteration_seconds": [round(v, 4) for v in per_iter]})
                return secs

            secs_probe = one(n_probe)
            # the mdot leg gets a sixth of the budget; the rest goes to up to three runs at the largest size that fits
            per_elem = secs_probe / n_probe
            n_big = n // 4 if n >= 20_000_000 else n
            left = 0.8 * budget_s - (time.time() - t_start)
            n_fit, reps = n_probe, 0
            for want in (3, 2, 1):
                cand = int(min(n_big, left / (want * per_elem * 1.15))) if per_elem > 0 else n_big
                if nwcon > 0:
                    cand -= cand % (nw * ranks)
                if cand >= 2 * n_probe:
                    n_fit, reps = cand, want
                    break
            for _ in range(reps):
                one(n_fit)
            best_n = runs[-1]["n"]
            same = [r for r in runs if r["n"] == best_n]
            # --cpu-full-size (builder runs only, never the driver's default): ONE run of the reference at the workload's
            # own n, so that the linear scaling of the sampled rate rests on a measurement.  Guarded by
            # the host's free memory: the reference holds about 30 + 2 c + 4 k design-sized vectors.
            full = None
            if full_size and n > best_n:
                kcols = qn_size * (2 if qn == "bfgs" else 1)
                need = 8.0 * n * (30 + 2 * ncon + 4 * kcols)
                avail = 0.0
                try:
                    with open("/proc/meminfo") as f:
                        for ln in f:
                            if ln.startswith("MemAvailable:"):
                                avail = 1024.0 * float(ln.split()[1])
                except Exception:
                    pass
                if avail > 1.3 * need:
                    nf, sf, pf, wf, ef = run_reference(drv, mpiexec, ranks, n, ncon, K, qn, qn_size, problem, nwcon, nw, 1800)
                    if nf > 0:
                        full = {"n": n, "iterations": nf, "seconds": sf, "wall_s": wf, "whole_run_it_per_s": nf / sf,
                                "steady_state_it_per_s": steady(pf), "iteration_seconds": [round(v, 4) for v in pf],
                                "host_mem_available_GB": avail * 1e-9, "estimated_need_GB": need * 1e-9}
                    else:
                        full = {"error": ef}
                else:
                    full = {"skipped": "host memory: %.0f GB available, about %.0f GB needed" % (avail * 1e-9, need * 1e-9)}
            best = max(same, key=lambda r: r["whole_run_it_per_s"])
            best_ss = max((r["steady_state_it_per_s"] for r in same if r["steady_state_it_per_s"]), default=None)
            scale = best_n / float(n)
            kw = [min(i, qn_size) * (2 if qn == "bfgs" else 1) for i in range(best["iterations"])]
            traffic = sum(reference_traffic_bytes(best_n, ncon, k) for k in kw)
            # ParOptVec::mdot alone (the headline kernel's CPU counterpart) at the workload's n when it fits
            mdot = {}
            n_md = n
            t_md = time.time()
            for nv in (8, 32, 40):
                if time.time() - t_start > budget_s + 20.0:
                    break
                try:
                    sec = run_reference_mdot(drv, mpiexec, ranks, n_md, nv, 3, 300)
                except Exception:  # pragma: no cover
                    sec = None
                if sec:
                    mdot["nvecs_%d" % nv] = {"ms": 1e3 * sec, "n": n_md,
                                             "algorithmic_GBps": 8.0 * (nv + 1) * n_md / sec * 1e-9}
            res = {"value": (best_ss if best_ss else best["whole_run_it_per_s"]) * scale,
                   "unit": "IP iterations/s", "cores": ranks, "kind": "reference",
                   "steady_state": best_ss is not None,
                   "whole_run_it_per_s": best["whole_run_it_per_s"] * scale,
                   "steady_state_it_per_s": (best_ss * scale) if best_ss else None,
                   "iterations_per_run": best["iterations"], "full_memory_iterations": max(0, best["iterations"] - kfull),
                   "runs_at_sample_n": len(same), "sample_n": best_n,
                   "seconds_per_iteration_at_sample_n": best["seconds"] / best["iterations"],
                   "implied_host_GBps": traffic / best["seconds"] * 1e-9,
                   "mdot_ms": mdot, "mdot_seconds_spent": time.time() - t_md,
                   "host": cpus, "wall_s_incl_launch": time.time() - t_start, "probes": runs,
                   "full_size_run": full,
                   "sample": "unmodified reference (%d MPICH ranks x sequential MKL; ranks = cores this process may use: "
                             "affinity %d, cgroup quota %s, os.cpu_count %d), same problem at n=%d, K = %d major iterations "
                             "from a cold quasi-Newton memory (BASELINE.md section 4), optimize() only; `value` = the "
                             "steady-state rate (iterations %d.. with full memory, per-iteration stamps of the driver), "
                             "best of %d run(s) at this n; every reference pass is O(n) and memory-bound, rates are "
                             "scaled by %d/%d to the workload's n; mdot_ms = ParOptBasicVec::mdot alone at n=%d; "
                             "implied_host_GBps = the reference sequence's traffic model (SURVEY 3.4) / seconds" % (
                                 ranks, cpus["affinity"], cpus["cgroup_quota_cpus"], cpus["os_cpu_count"], best_n, K,
                                 kfull, len(same), best_n, n, n_md)}
            # a full-size measurement on record (builder run with --cpu-full-size, committed under profiles/): what the
            # linear scaling of the sampled rate is worth at the workload's own n
            try:
                rec_path = os.path.join(ROOT, "profiles", "r05_bench_c3_cpu_full_size.json")
                with open(rec_path) as f:
                    rec = json.loads(f.readline())
                rc, fs = rec["config"], rec["cpu_baseline"]["full_size_run"]
                if (rc["n_global"], rc["ncon"], rc["nwcon"], rc["qn"]) == (n, ncon, nwcon, qn) and fs and "seconds" in fs:
                    res["full_size_on_record"] = {
                        "file": "profiles/r05_bench_c3_cpu_full_size.json", "n": fs["n"], "cores": rec["cpu_baseline"]["cores"],
                        "steady_state_it_per_s": fs["steady_state_it_per_s"], "whole_run_it_per_s": fs["whole_run_it_per_s"],
                        "scaled_sample_of_that_run_it_per_s": rec["cpu_baseline"]["steady_state_it_per_s"],
                        "note": "ONE run of the unmodified reference at the workload's own n on a builder box (K = 18): the "
                                "rate scaled linearly from the n / 4 sample of the same box was 2.1 x higher -- the host's "
                                "memory system does not scale linearly to 45 GB of vectors; `value` above is the scaled "
                                "sample of THIS box and overestimates the CPU by about that factor"}
            except Exception:
                pass
            if (cpus.get("loadavg_1min") or 0.0) > 0.75 * ranks:
                res["note"] = ("the host's 1-minute load average (%.1f) was already at the level of the %d granted cores "
                               "before the ranks started: other jobs share the cores' memory system, compare "
                               "implied_host_GBps between runs" % (cpus["loadavg_1min"], ranks))
            return res
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
            "steady_state": False, "host": cpus,
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
    ap.add_argument("--boundary", type=str, default="both", choices=["builtin", "facade", "both"],
                    help="builtin: the library's own SeparableProblem; facade: the same workload as a user's "
                         "ParOptProblem subclass outside the library (examples/random_convex_amd.cpp), which then "
                         "is `value`; both (default): `value` from the built-in, the facade measurement under "
                         "`boundary`")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-budget", type=float, default=60.0, help="seconds the CPU baseline leg may take")
    ap.add_argument("--cpu-full-size", action="store_true",
                    help="also run the reference ONCE at the workload's own n (minutes; builder runs only)")
    ap.add_argument("--cpu-iters", type=int, default=0,
                    help="major iterations of a CPU baseline run (default and minimum: qn_size + 8, so that at least "
                         "six run with full quasi-Newton memory)")
    ap.add_argument("--skip-extension-variant", action="store_true",
                    help="do not also measure the linear-constraint declaration (API extension) beside the headline")
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


_REAL_STDOUT = None


def claim_stdout():
    """stdout of a rank carries the ONE JSON line and nothing else: libraries loaded below write there too (RCCL prints
    its version banner to the C stdout of every rank that creates a communicator, buffered until the process exits --
    i.e. BEHIND the line), so file descriptor 1 is pointed at stderr for the run and the line goes to the saved one."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit_line(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _REAL_STDOUT is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        os.write(_REAL_STDOUT, data)


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
        emit_line({"metric": "stub", "n_gpus": seen, "steps": a.steps, "warmup": a.warmup, "stub": True,
                   "gpus_arg": a.gpus})
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

    claim_stdout()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # the host driver of the pool supports dmabuf IPC only (RCCL's buffer exchange between the ranks' processes)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
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
            # native ncclAllReduce / ncclAllGather on the solver's own stream; po_ctx_comm_init_rccl runs
            # known-answer collectives (sum of rank+1, rank order, SUM/MIN/MAX) before it returns
            ctx.init_rccl_from_torch()
            comm_kind = "rccl"
        except Exception as e:
            if not a.allow_fallback:
                log("native RCCL communicator failed (%r); refusing to measure a fallback (--allow-fallback)" % (e,))
                sys.exit(3)
            log("native RCCL communicator failed (%r); falling back to torch.distributed all_gather" % (e,))
            ctx.init_callback_from_torch(device=torch.device("cuda", local_rank))
            comm_kind = "torch.distributed(nccl) callback"
    forced_rccl = world == 1 and os.environ.get("PAROPT_AMD_FORCE_RCCL", "0") == "1"
    if forced_rccl:
        # One rank, but every reduction goes through the REAL RCCL path the multi-GPU runs take: ncclAllReduce /
        # ncclAllGather on the solver's stream, the publish kernel and the polled completion flag (context.cpp).  Run at
        # the per-rank size of an N-GPU job this measures what one rank of that job does, collectives included, except
        # the wire time of the <= 8.5 KB payloads (tools/per_rank_sizes.sh).
        import ctypes as C

        from paropt_amd.lib import check, lib

        buf = (C.c_char * 128)()
        check(lib.po_rccl_unique_id(buf))
        check(lib.po_ctx_comm_init_rccl(ctx.handle, 0, 1, buf))
        comm_kind = "rccl (forced single-rank communicator)"
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
    opts = {"qn_type": a.qn, "qn_subspace_size": a.qn_size, "abs_res_tol": 1e-30,
            "start_affine_multiplier_min": 0.01, "max_major_iters": W + K, "write_output_frequency": 0}

    def measure(prob, own_bytes=None):
        """R optimize() runs on `prob`; returns (median seconds of the K timed iterations, sorted times, details of
        the median run, all runs)."""
        ip = pa.InteriorPoint(prob, opts)
        ip.setCallbackTiming(True)  # user_eval_ms_per_iter below (event records around the problem's callbacks)
        stamp = {}

        def cb(k):
            if k == W:
                barrier_sync()
                stamp["counters0"] = ctx.counters()
                stamp["bytes0"] = ctx.algorithmic_bytes()
                stamp["own0"] = own_bytes() if own_bytes else 0.0
                ctx.time_mdot(a.ncon)  # HIP events around every mdot<ncon> launch of the timed region
                ctx.time_wgram(True)
                stamp["t0"] = time.perf_counter()

        ip.setIterationCallback(cb)

        def one_run():
            ip.resetQuasiNewtonHessian()
            barrier_sync()
            ip.optimize()
            barrier_sync()
            t1 = time.perf_counter()
            elapsed = t1 - stamp["t0"]
            red1, lau1 = ctx.counters()
            by1 = ctx.algorithmic_bytes()
            own1 = own_bytes() if own_bytes else 0.0
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
            own = own1 - stamp["own0"]
            return elapsed, dict(red=(red1 - stamp["counters0"][0]) / float(K),
                                 launches=(lau1 - stamp["counters0"][1]) / float(K), mdot_ms=mdot_ms, mdot_n=mdot_n,
                                 wgram=wg, niter=niter, neval=neval, phases=ip.getPhaseTimes(),
                                 bytes_total=(by1[0] - stamp["bytes0"][0] + own) / float(K),
                                 bytes_user=(by1[1] - stamp["bytes0"][1] + own) / float(K))

        runs = [one_run() for _ in range(max(1, a.repeats))]
        times = sorted(r[0] for r in runs)
        med = statistics.median(times)
        det = min(runs, key=lambda r: abs(r[0] - med))[1]
        ip.setIterationCallback(lambda k: None)  # break the cycle ip -> callback -> ip: the solver's vectors
        del ip                                   # (tens of GB at the full size) are freed before the next variant
        gc.collect()
        return med, times, det, runs

    def summary(v):
        """Per-variant numbers of the line: rate, host syncs, launches and the iteration-level roofline."""
        med, times, det, runs = v
        ms = 1e3 * med / K
        user_ms = 1e3 * det["phases"].get("user_eval", 0.0) / det["niter"]
        lib_bytes = det["bytes_total"] - det["bytes_user"]
        lib_ms = ms - user_ms
        return {"value": K / med, "ms_per_step": ms, "ms_per_step_min": 1e3 * times[0] / K,
                "ms_per_step_max": 1e3 * times[-1] / K, "user_eval_ms_per_iter": user_ms,
                "host_syncs_per_iter": det["red"], "launches_per_iter": det["launches"],
                "iteration_bytes": det["bytes_total"], "iteration_bytes_user_callbacks": det["bytes_user"],
                "iteration_frac": det["bytes_total"] / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS,
                "iteration_frac_excl_user_callbacks":
                    (lib_bytes / (lib_ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS) if lib_ms > 0 else None}

    # ---- the built-in problem: reference contract first (headline), the API extension beside it ----
    variants = {}
    prob = pa.SeparableProblem(ctx, a.problem, a.n, a.ncon, 0)
    nl, offset = prob.nvars, prob.offset
    if a.nwcon > 0:
        prob.setWeighting(a.nwcon, a.nw, 0, 0)
    if a.boundary != "facade":
        prob.setLinearConstraints(False)
        variants["jacobian_rewritten_every_gradient_call"] = measure(prob)
        if not a.skip_extension_variant:
            prob.setLinearConstraints(True)
            variants["linear_constraints_declared__API_EXTENSION"] = measure(prob)
            prob.setLinearConstraints(False)
    del prob
    gc.collect()

    # ---- the same workload through the user-side boundary (facade) ----
    boundary = None
    user_lib = USER_LIB if a.nwcon == 0 else WEIGHTING_LIB
    facade_ok = a.problem == "convex" and os.path.exists(user_lib)
    if a.boundary in ("facade", "both"):
        if not facade_ok:
            msg = "the facade workloads are configs 3 and 4 (convex) and need %s" % user_lib
            if a.boundary == "facade":
                log("error: " + msg)
                sys.exit(5)
            log("skipping the facade measurement: " + msg)
        elif a.nwcon > 0:
            # config 4: the weighting constraints reach the library ONLY through the reference's ParOptSparseProblem
            # interface (setSparseJacobianData pattern + the entries the gradient callback writes); the library
            # recognises the grouped pattern and runs its fused group kernels
            up = pa.UserLibraryProblem(ctx, user_lib, a.n, a.ncon, prefix="wt", nwcon=a.nwcon, nw=a.nw)
            fac = {"reference_semantics": measure(up, lambda: up.ownKernelBytes(True))}
            up.close()
            boundary = {"what": "config 4 as a USER ParOptSparseProblem subclass on include/ParOptAMD.hpp, compiled "
                                "outside libparopt_amd.so with its own HIP kernels (examples/weighting_amd.cpp): the "
                                "weighting constraints are handed over as the CSR pattern of setSparseJacobianData "
                                "(src/ParOptProblem.h:306-312) and the Jacobian entries written by "
                                "evalSparseObjConGradient; the library recognises the grouped pattern (checked on the "
                                "device after every gradient evaluation) and takes the fused group kernels; "
                                "evalObjConGradient rewrites the dense Jacobian and all sparse entries at every call",
                        **{k: summary(v) for k, v in fac.items()}}
            if a.boundary == "facade":
                variants = {"facade_reference_semantics": fac["reference_semantics"]}
        else:
            up = pa.UserLibraryProblem(ctx, USER_LIB, a.n, a.ncon)
            fac = {"reference_semantics": measure(up, lambda: up.ownKernelBytes(True))}
            up.setDeferredReductions(True)
            fac["deferred_reductions__API_EXTENSION"] = measure(up, lambda: up.ownKernelBytes(True))
            up.close()
            boundary = {"what": "the workload as a USER ParOptProblem subclass on include/ParOptAMD.hpp, compiled outside "
                                "libparopt_amd.so with its own HIP kernels (examples/random_convex_amd.cpp); "
                                "evalObjConGradient rewrites the Jacobian at every call; reference_semantics: every "
                                "reduction of the callbacks returns its value immediately; deferred_reductions: the "
                                "opt-in of po_problem_set_deferred_reductions",
                        **{k: summary(v) for k, v in fac.items()}}
            if a.boundary == "facade":
                variants = {"facade_reference_semantics": fac["reference_semantics"],
                            "facade_deferred_reductions__API_EXTENSION": fac["deferred_reductions__API_EXTENSION"]}
    head = next(iter(variants))
    elapsed, times, det, runs = variants[head]
    all_runs = [r for v in variants.values() for r in v[3]]

    # ---- roofline of the headline kernel, measured live with HIP events on the context stream ----
    x = pa.PVec(ctx, nl).fill_hash(1, 10, offset, 2.0, -1.0)
    # the dense-constraint panel the solver itself streams: fresh hash vectors of the same shape
    V = [pa.PVec(ctx, nl).fill_hash(1, 20 + j, offset, 2.0, -1.0) for j in range(a.ncon)]
    ms_isolated, _ = pa.bench_mdot(x, V, 20)
    # same-run stream ceilings: what a read-only stream and a copy reach on this box right now
    ms_ro = pa.bench_stream(x, V[0], 0, 20)
    ms_cp = pa.bench_stream(x, V[0], 1, 20)
    stream = {"read_only_GBps": 16.0 * nl / (ms_ro * 1e-3) * 1e-9, "copy_GBps": 16.0 * nl / (ms_cp * 1e-3) * 1e-9,
              "what": "x.y over two %d-element vectors (read-only) and y <- x (copy), 20 launches each, "
                      "HIP events on the solver's stream, same process as the timed run" % nl}
    # the figure of record: the mdot<ncon> launches issued in the timed regions (the constraint evaluations of the
    # line search: the problem's own ParOptVec::mdot calls), HIP events on the launch stream
    mdot_ms_run = sum(r[1]["mdot_ms"] for r in all_runs)
    mdot_launches_run = sum(r[1]["mdot_n"] for r in all_runs)
    ms = mdot_ms_run / mdot_launches_run if mdot_launches_run > 0 else ms_isolated
    alg_bytes = 8.0 * (a.ncon + 1) * nl
    achieved = alg_bytes / (ms * 1e-3) * 1e-9
    traffic, traffic_src = None, None
    # PMC bytes per launch from committed rocprofv3 passes (NOT measured in this run) of this exact shape: round 4's
    # collection of the configuration (per-kernel FETCH_SIZE / WRITE_SIZE), else the older files
    # (the newest collection on record first: tools/collect_r06.sh, then the earlier rounds')
    for fn in ("r06_pmc_c3.json", "r05_pmc_c3_final_build.json", "r05_pmc_c3.json", "r04_pmc_c3.json"):
        try:
            if traffic is None and nl == 50_000_000 and a.ncon == 32:
                k = json.load(open(os.path.join(ROOT, "profiles", fn)))["void po::mdot_kernel<32>"]
                traffic = k["hbm_read_bytes_corrected"] + k["hbm_write_bytes"]
                traffic_src = ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over the same command, "
                               "%d launches; not collected in this run)" % (fn, k["FETCH_SIZE"]["dispatches"]))
        except Exception:
            pass
    for fn in ("r03_pmc_hbm_traffic.json", "r02_pmc_hbm_traffic.json", "r01_pmc_hbm_traffic.json"):
        if traffic is not None:
            break
        try:
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
        wms = sum(r[1]["wgram"][w][0] for r in all_runs)
        wn = sum(r[1]["wgram"][w][1] for r in all_runs)
        if wn > 0:
            cols = all_runs[0][1]["wgram"][w][2]
            wbytes = sum(r[1]["wgram"][w][3] for r in all_runs) / wn
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

    # ---- N > 1: what one reduction exchange costs on this communicator (the iteration pays host_syncs_per_iter
    # of them), in the two forms the solver uses ----
    collective = None
    if world > 1 or forced_rccl:
        collective = {"allreduce_2628_doubles": ctx.bench_collective(2628, True, 50),
                      "allgather_16_doubles": ctx.bench_collective(16, False, 50),
                      "what": "final-stage payload -> collective -> device-to-host copy -> host sync, median of 50 "
                              "(po_ctx_bench_collective); 2628 doubles = the Gram payload at c + k = 72"}

    if rank == 0:
        cpu = None
        if not a.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(a.n, a.ncon, a.cpu_iters, log, a.nwcon, a.nw, a.qn, a.qn_size, a.problem, a.cpu_budget,
                               a.cpu_full_size)
        niter = det["niter"]
        kind, nred, ngat = ctx.comm_info()
        head_sum = summary(variants[head])
        k_full = a.qn_size * (2 if a.qn == "bfgs" else 1)
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
                       "n_global": a.n, "n_local": nl, "ncon": a.ncon, "nwcon": a.nwcon, "qn": a.qn,
                       "evals_per_iter": (det["neval"] - 1) / float(niter),
                       "collective": comm_kind, "reductions_per_iter": det["red"],
                       "launches_per_iter": det["launches"],
                       "rccl_allreduce_calls": nred, "rccl_allgather_calls": ngat,
                       "launcher": "torchrun rank" if in_job else "direct",
                       "headline_variant": head,
                       "problem_contract": "reference (src/ParOptProblem.h:146-189): evalObjConGradient rewrites the "
                                           "whole constraint Jacobian at every call; `value` uses no API extension"},
            # iteration-level roofline: algorithmic bytes of every n-sized launch of the timed
            # iterations (each operand stream of a launch counted once: po_ctx_algorithmic_bytes; for the facade the
            # user's own kernels by their formula) / step time / HBM peak
            "iteration_bytes": head_sum["iteration_bytes"],
            "iteration_bytes_user_callbacks": head_sum["iteration_bytes_user_callbacks"],
            "iteration_frac": head_sum["iteration_frac"],
            "iteration_frac_basis": "rank-local algorithmic bytes (n_local = %d of n = %d) / max-over-ranks step time "
                                    "/ 8 TB/s: a per-GPU fraction at every N" % (nl, a.n),
            "iteration_frac_excl_user_callbacks": head_sum["iteration_frac_excl_user_callbacks"],
            "iteration_bytes_models": {
                "survey_8d_fused_model_8n(165+15c+11k)": 8.0 * nl * (165 + 15 * a.ncon + 11 * k_full),
                "reference_sequence_8n(332+48c+c^2+45k+5ck+2k^2)": reference_traffic_bytes(nl, a.ncon, k_full),
                "note": "the build needs fewer bytes than SURVEY 8d's fused model assumed (3 passes over the panel "
                        "per iteration); `iteration_bytes` is what its launches actually stream"},
            "variants": {k: summary(v) for k, v in variants.items()},
            "boundary": boundary,
            "roofline": roofline,
            "collective_us": collective,
            # polled completions of the reductions and those that ran out of their bounded spin (po_ctx_sync_counters)
            "sync_counters": ctx.sync_counters(),
            "cpu_baseline": cpu,
            "phase_ms_per_iter": {k: 1e3 * v / niter for k, v in det["phases"].items()},
            "user_eval_ms_per_iter": head_sum["user_eval_ms_per_iter"],
        }
        emit_line(res)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
