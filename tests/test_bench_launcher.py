"""bench.py --gpus N: the N-rank job is started as a child process (torchrun) by bench.py itself when it is
not already running under one; the line it prints reports the ranks the job really had."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env_extra, timeout):
    env = dict(os.environ, **env_extra)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, env=env, capture_output=True,
                         text=True, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    return out, lines


def test_self_launch_plumbing_cpu():
    """No GPU: PAROPT_BENCH_STUB=1 replaces the solver by a gloo all-reduce, everything else (argument relay,
    torchrun child, rank-0 line relay, exit code) is the real launcher."""
    out, lines = _run(["--gpus", "2", "--steps", "3", "--warmup", "2"], {"PAROPT_BENCH_STUB": "1"}, 300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2 and r["gpus_arg"] == 2 and r["steps"] == 3 and r["warmup"] == 2


def test_single_rank_does_not_spawn_cpu():
    out, lines = _run(["--gpus", "1", "--steps", "1"], {"PAROPT_BENCH_STUB": "1"}, 120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert json.loads(lines[0])["n_gpus"] == 1
    assert "launching" not in out.stderr


def test_child_failure_is_reported_cpu():
    """A job whose ranks fail must not look like a result: non-zero exit, no line."""
    out, lines = _run(["--gpus", "2", "--steps", "1", "--problem", "nonsense"], {"PAROPT_BENCH_STUB": "1"}, 300)
    assert out.returncode != 0
    assert not lines


@pytest.mark.gpu
def test_bench_gpus2_shared_gpu():
    """`python bench.py --gpus 2` (the shape of the driver's command) on a 1-GPU box: both ranks share GPU 0 and
    reduce through the host-callback communicator; the line must say n_gpus == 2 and carry the contract fields."""
    out, lines = _run(["--gpus", "2", "--nglobal", "2000000", "--steps", "3", "--warmup", "3", "--repeats", "2",
                       "--no-cpu-baseline", "--qn-size", "3", "--boundary", "builtin"], {"PAROPT_BENCH_SHARE_GPU": "1"},
                      900)
    assert out.returncode == 0, out.stderr[-3000:]
    assert len(lines) == 1, out.stdout
    r = json.loads(lines[0])
    assert r["n_gpus"] == 2
    assert r["steps"] == 3 and r["warmup"] == 5  # clamped to qn_size + 2
    assert r["repeats"] == 2 and r["ms_per_step_min"] <= r["ms_per_step"] <= r["ms_per_step_max"]
    assert r["config"]["collective"].startswith("gloo callback")
    assert list(r["variants"]) == ["jacobian_rewritten_every_gradient_call",
                                   "linear_constraints_declared__API_EXTENSION"]
    assert r["config"]["headline_variant"] == "jacobian_rewritten_every_gradient_call"
    assert r["config"]["n_local"] == 1000000
    assert r["roofline"]["stream_ceiling"]["read_only_GBps"] > 0
    assert r["roofline"]["second"] is not None and r["roofline"]["second"]["hbm"]["frac"] > 0
    assert r["user_eval_ms_per_iter"] > 0


@pytest.mark.gpu
def test_bench_single_gpu_line():
    out, lines = _run(["--nglobal", "1000000", "--steps", "3", "--warmup", "3", "--repeats", "1",
                       "--no-cpu-baseline", "--qn-size", "3"], {}, 900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads(lines[0])
    assert r["n_gpus"] == 1 and r["metric"].startswith("IP iterations/sec") and "n=1M vars m=32" in r["metric"]
    assert r["roofline"]["traffic"] is None  # the committed PMC profile is for n = 50 M only
    v = r["variants"]
    # `value` is the reference-contract variant (Jacobian rewritten at every gradient call), never the extension
    head = v["jacobian_rewritten_every_gradient_call"]
    assert r["config"]["headline_variant"] == "jacobian_rewritten_every_gradient_call"
    assert abs(r["value"] - head["value"]) < 1e-9 * r["value"] and r["ms_per_step"] == head["ms_per_step"]
    ext = v["linear_constraints_declared__API_EXTENSION"]
    assert ext["user_eval_ms_per_iter"] < head["user_eval_ms_per_iter"]
    # iteration-level roofline: algorithmic bytes per iteration, the user callbacks' share, and the fractions
    assert r["iteration_bytes"] > r["iteration_bytes_user_callbacks"] > 0
    assert 0 < r["iteration_frac"] < 1 and 0 < r["iteration_frac_excl_user_callbacks"] < 1
    # one iteration streams the 42-column panel three times plus the problem's own passes: between 150 and 500
    # doubles per design variable at c = 32 (the reference's sequence: ~5200)
    assert 150 < r["iteration_bytes"] / (8 * 1_000_000) < 500
    assert ext["iteration_bytes"] < head["iteration_bytes"]
    # the same workload through the user-side boundary (examples/random_convex_amd.cpp), both reduction modes
    b = r["boundary"]
    assert b is not None
    fr, fd = b["reference_semantics"], b["deferred_reductions__API_EXTENSION"]
    assert fd["host_syncs_per_iter"] < fr["host_syncs_per_iter"]
    assert fd["host_syncs_per_iter"] <= head["host_syncs_per_iter"] + 0.5
    assert abs(fr["iteration_bytes"] - head["iteration_bytes"]) < 0.05 * head["iteration_bytes"]


@pytest.mark.gpu
def test_bench_boundary_facade_is_value():
    out, lines = _run(["--nglobal", "1000000", "--steps", "3", "--warmup", "3", "--repeats", "1",
                       "--no-cpu-baseline", "--qn-size", "3", "--boundary", "facade"], {}, 900)
    assert out.returncode == 0, out.stderr[-3000:]
    r = json.loads(lines[0])
    assert r["config"]["headline_variant"] == "facade_reference_semantics"
    assert abs(r["value"] - r["boundary"]["reference_semantics"]["value"]) < 1e-9 * r["value"]


@pytest.mark.gpu
def test_torchrun_child_at_n1_equals_the_in_process_path():
    """one rank started the way the driver starts N ranks (python -m torch.distributed.run
    --nproc-per-node 1 ... bench.py --gpus 1: WORLD_SIZE = 1, RCCL process group of one) measures what the in-process
    N = 1 path measures -- same counters per iteration, `value` within 2 % -- and the line's `iteration_frac` is built
    from the rank-local bytes and the max-over-ranks time."""
    import socket

    args = ["--gpus", "1", "--nglobal", "16000000", "--steps", "10", "--warmup", "12", "--repeats", "5",
            "--no-cpu-baseline", "--skip-extension-variant", "--boundary", "builtin"]

    def run_direct():
        out, lines = _run(args, {}, 900)
        assert out.returncode == 0, out.stderr[-3000:]
        return json.loads(lines[0])

    def run_child():
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
               "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py")] + args
        res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert res.returncode == 0, res.stderr[-3000:]
        return json.loads([ln for ln in res.stdout.splitlines() if ln.startswith("{")][0])

    # the in-process path before AND after the child (the first process of a fresh box runs a few per cent slower while
    # clocks and caches settle), the fastest repeat of each run (`ms_per_step_min`): launcher overhead, not noise
    direct = run_direct()
    child = run_child()
    direct2 = run_direct()
    assert child["n_gpus"] == 1 and child["config"]["launcher"] == "torchrun rank"
    assert child["config"]["reductions_per_iter"] == direct["config"]["reductions_per_iter"]
    assert child["config"]["launches_per_iter"] == direct["config"]["launches_per_iter"]
    assert child["iteration_bytes"] == direct["iteration_bytes"]
    ref = [direct["ms_per_step_min"], direct2["ms_per_step_min"]]
    rel = min(abs(child["ms_per_step_min"] / r - 1.0) for r in ref)
    assert rel <= 0.02, (child["ms_per_step_min"], ref)
    assert "n_local" in child["iteration_frac_basis"]
    frac = child["iteration_bytes"] / (child["ms_per_step"] * 1e-3) * 1e-9 / 8000.0
    assert abs(frac - child["iteration_frac"]) <= 1e-9


def test_host_cpu_budget_respects_affinity_cpu():
    """The CPU baseline sizes its MPI job from the cores the process may really use (affinity, cgroup quota)."""
    sys.path.insert(0, ROOT)
    import bench

    info = bench.host_cpu_budget()
    assert 1 <= info["usable"] <= info["affinity"] <= max(info["os_cpu_count"], info["affinity"])
    if info["cgroup_quota_cpus"] is not None:
        assert info["usable"] <= max(1, int(info["cgroup_quota_cpus"]))
    # traffic model of the reference's own sequence: 5200 doubles per variable at c = 32, k = 10
    assert abs(bench.reference_traffic_bytes(1, 32, 10) / 8 - (332 + 48 * 32 + 32 * 32 + 450 + 1600 + 200)) < 1e-9
