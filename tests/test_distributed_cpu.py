"""
World-size-2 checks on the CPU (gloo): the 1-D row-block sharding of the design vector
 and the rule that every synthetic array is a pure function of the GLOBAL index.
The sharded oracle (reductions through torch.distributed) must follow the single-rank trajectory:
integer bookkeeping identical, scalars to 1e-9 -- the same statement the reference satisfies across
MPI rank counts (tests/golden/ip_convex_n2000_c32_bfgs vs _r2).
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from oracle import paropt_oracle as po


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, problem, n, c, opts, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = po.TorchComm()
    prob = po.SepProblem(problem, n, c, comm=comm)
    ip = po.InteriorPoint(prob, opts, comm=comm)
    snaps = []
    ip.hook = lambda s, k: snaps.append((s.niter, s.neval, s.ngeval, s.fobj, s.barrier_param,
                                         float(s.ops.norm(s.vars.x)), len(s.qn.Z)))
    ip.optimize()
    # gather the shards of x to rank 0 for an element-wise comparison
    xs = [None] * world
    dist.all_gather_object(xs, (prob.offset, ip.vars.x.copy()))
    if rank == 0:
        q.put((snaps, [t["info"] for t in ip.trace], np.concatenate([x for _, x in sorted(xs, key=lambda t: t[0])])))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("problem,n,c,qn", [("quadratic", 1001, 3, "bfgs"), ("convex", 500, 5, "sr1")])
def test_sharded_oracle_matches_single_rank(problem, n, c, qn):
    opts = {"qn_type": qn, "qn_subspace_size": 5, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01,
            "max_major_iters": 15 if qn == "bfgs" else 8}
    ref = po.InteriorPoint(po.SepProblem(problem, n, c), opts)
    rsn = []
    ref.hook = lambda s, k: rsn.append((s.niter, s.neval, s.ngeval, s.fobj, s.barrier_param,
                                        float(s.ops.norm(s.vars.x)), len(s.qn.Z)))
    ref.optimize()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, problem, n, c, opts, q)) for r in range(2)]
    for p in procs:
        p.start()
    snaps, infos, x = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(snaps) == len(rsn)
    for a, b in zip(snaps, rsn):
        assert a[:3] == b[:3] and a[6] == b[6]          # counters, quasi-Newton size: exact
        np.testing.assert_allclose(a[3:6], b[3:6], rtol=1e-9)
    assert infos == [t["info"] for t in ref.trace]
    np.testing.assert_allclose(x, ref.vars.x, rtol=0, atol=1e-9)


def test_shard_partition_properties():
    for n in (1, 7, 50_000_000, 50_000_003):
        for size in (1, 2, 4, 8):
            parts = [po.shard(n, r, size) for r in range(size)]
            assert sum(p[0] for p in parts) == n
            off = 0
            for nl, o in parts:
                assert o == off
                off += nl
            assert max(p[0] for p in parts) - min(p[0] for p in parts) <= 1


def test_hash_data_is_sharding_invariant():
    full = po.u01(7, 123, np.arange(1000, dtype=np.uint64))
    for size in (2, 3, 8):
        pieces = []
        for r in range(size):
            nl, off = po.shard(1000, r, size)
            pieces.append(po.u01(7, 123, np.arange(off, off + nl, dtype=np.uint64)))
        np.testing.assert_array_equal(np.concatenate(pieces), full)


def _worker_csr(rank, world, port, args, opts, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    comm = po.TorchComm()
    prob = po.SepProblem(args["problem"], args["n"], args["c"], comm=comm,
                         chain=(args["chain_span"], args.get("chain_stride", 1)))
    ip = po.InteriorPoint(prob, opts, comm=comm)
    snaps = []
    ip.hook = lambda s, k: snaps.append(s.snapshot())
    ip.optimize()
    if rank == 0:
        q.put(([(tuple(s["counters"]), s["fobj"], s["mu"], tuple(s["norms"]), tuple(s["wnorms"])) for s in snaps],
               (ip.niter, ip.neval, ip.ngeval), ip.fobj))
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_oracle_csr_constraints_match_reference_on_two_ranks():
    """Rank-local CSR sparse constraints (each rank owns the rows over its own variables and its own S):
    the sharded oracle against the trajectory the reference produced on two MPI ranks."""
    from conftest import ip_options_from_case, load_golden

    g, case = load_golden("ipcsr_convex_n240_c2_chain2_r2")
    opts = ip_options_from_case(case)
    opts.pop("write_output_frequency", None)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_csr, args=(r, 2, port, case["args"], opts, q)) for r in range(2)]
    for p in procs:
        p.start()
    snaps, counters, fobj = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nref = 1 + max(int(k[2:5]) for k in g if k.startswith("it") and k.endswith("/mu"))
    for k in range(min(25, nref, len(snaps))):
        pfx = "it%03d/" % k
        cnt, f, mu, norms, wnorms = snaps[k]
        np.testing.assert_array_equal(np.array(cnt), g[pfx + "counters"])
        assert abs(f - g[pfx + "fobj"][0]) <= 1e-7 * max(1.0, abs(g[pfx + "fobj"][0]))
        assert abs(mu - g[pfx + "mu"][0]) <= 1e-7 * abs(g[pfx + "mu"][0])
        np.testing.assert_allclose(norms, g[pfx + "norms"], rtol=1e-7)
        np.testing.assert_allclose(wnorms, g[pfx + "wnorms"], rtol=1e-6)
    np.testing.assert_array_equal(np.array(counters), g["final/counters"])
    assert abs(fobj - g["final/fobj"][0]) <= 1e-6 * max(1.0, abs(g["final/fobj"][0]))
