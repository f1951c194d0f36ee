"""
Two ranks sharing ONE GPU (the test box has a single MI355X): the sharded product path -- shard
offsets, global-index data, per-rank partial reductions, rank-ordered all-gather combine -- through
the host-callback communicator (gloo underneath).  RCCL itself needs one GPU per rank and is
exercised by bench.py on the multi-GPU node.
"""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import paropt_amd as pa

    ctx = pa.Context(0)
    ctx.init_callback_from_torch()
    assert ctx.rank_size() == (rank, world)
    n, c = 40003, 6
    prob = pa.SeparableProblem(ctx, "convex", n, c)
    # vector reductions over the sharded vector
    v = pa.PVec(ctx, prob.nvars).fill_hash(0, 10, prob.offset, 2.0, -1.0)
    w = pa.PVec(ctx, prob.nvars).fill_hash(0, 11, prob.offset, 2.0, -1.0)
    red = (v.dot(w), v.norm(), v.maxabs(), v.l1norm(), list(v.mdot([w, v])))
    opts = {"qn_type": "bfgs", "qn_subspace_size": 6, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01,
            "max_major_iters": 20, "write_output_frequency": 0}
    ip = pa.InteriorPoint(prob, opts)
    snaps = []
    ip.setIterationCallback(lambda k: snaps.append(ip.snapshot()))
    ip.optimize()
    x = ip.getOptimizedPoint()[0].to_numpy()
    xs = [None] * world
    dist.all_gather_object(xs, (prob.offset, x))
    if rank == 0:
        q.put((red, [(tuple(s["counters"]), s["qn_size"], s["fobj"], s["mu"], tuple(s["norms"])) for s in snaps],
               np.concatenate([a for _, a in sorted(xs, key=lambda t: t[0])])))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_match_single_rank():
    import paropt_amd as pa

    ctx = pa.Context(0)
    n, c = 40003, 6
    prob = pa.SeparableProblem(ctx, "convex", n, c)
    v = pa.PVec(ctx, n).fill_hash(0, 10, 0, 2.0, -1.0)
    w = pa.PVec(ctx, n).fill_hash(0, 11, 0, 2.0, -1.0)
    red1 = (v.dot(w), v.norm(), v.maxabs(), v.l1norm(), list(v.mdot([w, v])))
    opts = {"qn_type": "bfgs", "qn_subspace_size": 6, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01,
            "max_major_iters": 20, "write_output_frequency": 0}
    ip = pa.InteriorPoint(prob, opts)
    s1 = []
    ip.setIterationCallback(lambda k: s1.append(ip.snapshot()))
    ip.optimize()
    x1 = ip.getOptimizedPoint()[0].to_numpy()

    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    red2, s2, x2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    np.testing.assert_allclose(red2[0], red1[0], rtol=0, atol=1e-10)
    np.testing.assert_allclose(red2[1], red1[1], rtol=1e-13)
    assert red2[2] == red1[2]
    np.testing.assert_allclose(red2[3], red1[3], rtol=1e-13)
    np.testing.assert_allclose(red2[4], red1[4], rtol=0, atol=1e-9)
    assert len(s2) == len(s1)
    for a, b in zip(s2, s1):
        assert a[0] == tuple(b["counters"]) and a[1] == b["qn_size"]
        assert abs(a[2] - b["fobj"]) <= 1e-7 * max(1.0, abs(b["fobj"]))
        assert abs(a[3] - b["mu"]) <= 1e-7 * abs(b["mu"])
        np.testing.assert_allclose(a[4], b["norms"], rtol=1e-7)
    np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-7)


def _worker_w(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import paropt_amd as pa

    ctx = pa.Context(0)
    ctx.init_callback_from_torch()
    n, c, nw = 8000, 3, 20
    prob = pa.SeparableProblem(ctx, "convex", n, c).setWeighting(n // nw, nw, 0, 0)
    assert prob.nwcon == n // nw // world
    ip = pa.InteriorPoint(prob, W_OPTS)
    snaps = []
    ip.setIterationCallback(lambda k: snaps.append(ip.snapshot()))
    ip.optimize()
    x = ip.getOptimizedPoint()[0].to_numpy()
    zw = ip.getOptimizedSparse()[0].to_numpy()
    xs = [None] * world
    dist.all_gather_object(xs, (prob.offset, x, zw))
    if rank == 0:
        xs = sorted(xs, key=lambda t: t[0])
        q.put(([(tuple(s["counters"]), s["qn_size"], s["fobj"], s["mu"], tuple(s["norms"]), tuple(s["wnorms"]))
                for s in snaps], np.concatenate([a for _, a, _ in xs]), np.concatenate([b for _, _, b in xs])))
    dist.barrier()
    dist.destroy_process_group()


W_OPTS = {"qn_type": "bfgs", "qn_subspace_size": 6, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01,
          "starting_point_strategy": "affine_step", "penalty_gamma": 1000.0, "max_major_iters": 15,
          "write_output_frequency": 0}


def test_two_ranks_weighting_constraints_match_single_rank():
    """Sparse (weighting) constraints sharded over two ranks: each rank owns the groups of its own
    variables, every w-sized reduction goes through the same rank-ordered combine."""
    import paropt_amd as pa

    ctx = pa.Context(0)
    n, c, nw = 8000, 3, 20
    prob = pa.SeparableProblem(ctx, "convex", n, c).setWeighting(n // nw, nw, 0, 0)
    ip = pa.InteriorPoint(prob, W_OPTS)
    s1 = []
    ip.setIterationCallback(lambda k: s1.append(ip.snapshot()))
    ip.optimize()
    x1 = ip.getOptimizedPoint()[0].to_numpy()
    zw1 = ip.getOptimizedSparse()[0].to_numpy()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker_w, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    s2, x2, zw2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(s2) == len(s1)
    for a, b in zip(s2, s1):
        assert a[0] == tuple(b["counters"]) and a[1] == b["qn_size"]
        assert abs(a[2] - b["fobj"]) <= 1e-7 * max(1.0, abs(b["fobj"]))
        assert abs(a[3] - b["mu"]) <= 1e-6 * abs(b["mu"])
        np.testing.assert_allclose(a[4], b["norms"], rtol=1e-6)
        np.testing.assert_allclose(a[5], b["wnorms"], rtol=1e-6)
    np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-6)
    np.testing.assert_allclose(zw2, zw1, rtol=0, atol=1e-6 * max(1.0, np.abs(zw1).max()))


def _worker_csr(rank, world, port, q, args, opts):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import paropt_amd as pa

    ctx = pa.Context(0)
    ctx.init_callback_from_torch()
    prob = pa.SeparableProblem(ctx, args["problem"], args["n"], args["c"]).setChain(
        args["chain_span"], args.get("chain_stride", 1), args.get("chain_reverse", 0))
    ip = pa.InteriorPoint(prob, opts)
    snaps = []
    ip.setIterationCallback(lambda k: snaps.append(ip.snapshot()))
    ip.optimize()
    if rank == 0:
        q.put(([(tuple(s["counters"]), s["qn_size"], s["fobj"], s["mu"], tuple(s["norms"]), tuple(s["wnorms"]),
                 tuple(s["z"])) for s in snaps], tuple(ip.getIterationCounters()), ip.getObjective()[0]))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_csr_constraints_match_reference_golden():
    """The CSR form of the sparse constraints on two ranks (each rank factors its own S, no collective on the
    sparse path) against the trajectory the reference produced on two MPI ranks."""
    from conftest import ip_options_from_case, load_golden

    g, case = load_golden("ipcsr_convex_n240_c2_chain2_r2")
    assert case["ranks"] == 2
    opts = ip_options_from_case(case)
    opts["write_output_frequency"] = 0
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker_csr, args=(r, 2, port, q, case["args"], opts)) for r in range(2)]
    for p in procs:
        p.start()
    snaps, counters, fobj = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nref = 1 + max(int(k[2:5]) for k in g if k.startswith("it") and k.endswith("/mu"))
    ncmp = min(25, nref, len(snaps))
    assert ncmp >= min(25, nref)
    for k in range(ncmp):
        pfx = "it%03d/" % k
        cnt, qs, f, mu, norms, wnorms, z = snaps[k]
        np.testing.assert_array_equal(np.array(cnt), g[pfx + "counters"], err_msg="counters @%d" % k)
        assert qs == int(g[pfx + "qn_size"][0])
        assert abs(mu - g[pfx + "mu"][0]) <= 1e-6 * abs(g[pfx + "mu"][0])
        assert abs(f - g[pfx + "fobj"][0]) <= 1e-6 * max(1.0, abs(g[pfx + "fobj"][0]))
        np.testing.assert_allclose(norms, g[pfx + "norms"], rtol=1e-6)
        np.testing.assert_allclose(wnorms, g[pfx + "wnorms"], rtol=1e-6)
        np.testing.assert_allclose(z, g[pfx + "z"], rtol=1e-5, atol=1e-5 * max(1.0, np.abs(g[pfx + "z"]).max()))
    np.testing.assert_array_equal(np.array(counters), g["final/counters"])
    assert abs(fobj - g["final/fobj"][0]) <= 1e-6 * max(1.0, abs(g["final/fobj"][0]))


def _worker_tr(rank, world, port, q):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import paropt_amd as pa

    ctx = pa.Context(0)
    ctx.init_callback_from_torch()
    prob = pa.SeparableProblem(ctx, "quadratic", 6001, 3)
    tr = pa.TrustRegion(prob, TR_OPTS)
    tr.setEigenModelSynthetic(4, 0, 0, 2.0)
    rows = []
    tr.setIterationCallback(lambda i: rows.append(tr.getLastRow()) if i > 0 else None)
    tr.optimize()
    rows.append(tr.getLastRow())
    x = tr.getOptimizedPoint()[0].to_numpy()
    xs = [None] * world
    dist.all_gather_object(xs, (prob.offset, x))
    if rank == 0:
        q.put((rows, np.concatenate([a for _, a in sorted(xs, key=lambda t: t[0])])))
    dist.barrier()
    dist.destroy_process_group()


TR_OPTS = {"qn_subspace_size": 5, "tr_max_iterations": 8}


def test_two_ranks_trust_region_eigen_model_match_single_rank():
    """The trust-region driver with the compact eigenvalue model on a sharded design vector: the
    model directions are generated from global indices, every model evaluation is a sharded mdot."""
    import paropt_amd as pa

    ctx = pa.Context(0)
    prob = pa.SeparableProblem(ctx, "quadratic", 6001, 3)
    tr = pa.TrustRegion(prob, TR_OPTS)
    tr.setEigenModelSynthetic(4, 0, 0, 2.0)
    rows1 = []
    tr.setIterationCallback(lambda i: rows1.append(tr.getLastRow()) if i > 0 else None)
    tr.optimize()
    rows1.append(tr.getLastRow())
    x1 = tr.getOptimizedPoint()[0].to_numpy()
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker_tr, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    rows2, x2 = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert len(rows2) == len(rows1) == 8
    for (v2, t2), (v1, t1) in zip(rows2, rows1):
        assert t2 == t1  # iteration counts of both subproblem solves, accept/reject flags
        np.testing.assert_allclose(v2, v1, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(x2, x1, rtol=0, atol=1e-7)


def test_rccl_plumbing_single_rank(monkeypatch):
    """ncclGetUniqueId / ncclCommInitRank / ncclAllGather through the dlopen'ed librccl with a
    1-rank communicator on the only GPU of the test box: same results as the self communicator."""
    import ctypes as C

    import paropt_amd as pa
    from paropt_amd.lib import check, lib

    monkeypatch.setenv("PAROPT_AMD_FORCE_RCCL", "1")
    ctx = pa.Context(0)
    buf = (C.c_char * 128)()
    check(lib.po_rccl_unique_id(buf))
    assert any(b != b"\x00" for b in buf)
    check(lib.po_ctx_comm_init_rccl(ctx.handle, 0, 1, buf))
    ref = pa.Context(0)
    n = 100003
    outs = []
    for c in (ctx, ref):
        x = pa.PVec(c, n).fill_hash(0, 10, 0, 2.0, -1.0)
        V = [pa.PVec(c, n).fill_hash(0, 20 + j, 0, 2.0, -1.0) for j in range(5)]
        d = pa.PVec(c, n).fill_hash(0, 9, 0, 1.0, 0.5)
        outs.append((x.mdot(V), x.norm(), x.maxabs(), pa.wgram(d, V)))
        ip = pa.InteriorPoint(pa.SeparableProblem(c, "quadratic", 5000, 3),
                              {"max_major_iters": 10, "write_output_frequency": 0, "qn_subspace_size": 4})
        ip.optimize()
        outs[-1] += (ip.getObjective()[0],)
    for a, b in zip(*outs):
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b))
    # pure-sum payloads (dot, mdot, Gram) went through ncclAllReduce, mixed SUM/MIN/MAX ones through ncclAllGather
    kind, nred, ngat = ctx.comm_info()
    assert kind == 1 and nred > 10 and ngat > 10
    assert ref.comm_info() == (0, 0, 0)
    # po_ctx_comm_init_rccl ran its known-answer collectives (they are not counted above) and checked the library's
    # version against the ABI the hand-declared prototypes assume
    ver = C.c_int()
    check(lib.po_rccl_version(C.byref(ver)))
    assert ver.value >= 21000, ver.value
    lat = ctx.bench_collective(64, True, 10)
    assert lat["median_us"] > 0 and ctx.comm_info()[1] == nred  # the latency probe is not counted either
    np.testing.assert_array_equal(ctx.allreduce(np.array([3.0, 4.0]), "sum"), [3.0, 4.0])
    # round 5: behind the collective a one-workgroup kernel publishes the results into pinned host memory and raises
    # the completion flag; the host polled it for every exchange above and never ran out of its bounded spin
    sc = ctx.sync_counters()
    assert sc["flag_waits"] >= nred + ngat and sc["flag_timeouts"] == 0, sc
    assert sc["allreduces"] == ctx.comm_info()[1] and sc["allgathers"] == ctx.comm_info()[2]


@pytest.mark.parametrize("forced_rccl", [False, True])
def test_completion_flag_soak(monkeypatch, forced_rccl):
    """1e5 reductions (2e4 through the forced single-rank RCCL communicator: ncclAllReduce / ncclAllGather + publish
    kernel) finish through the polled completion flag: not one bounded spin runs out (po_ctx_sync_counters), every
    result equals the first one bit for bit, and batched reductions (one flag for several) are among them."""
    import ctypes as C

    import paropt_amd as pa
    from paropt_amd.lib import check, lib

    ctx = pa.Context(0)
    if forced_rccl:
        monkeypatch.setenv("PAROPT_AMD_FORCE_RCCL", "1")
        buf = (C.c_char * 128)()
        check(lib.po_rccl_unique_id(buf))
        check(lib.po_ctx_comm_init_rccl(ctx.handle, 0, 1, buf))
    n = 4099
    x = pa.PVec(ctx, n).fill_hash(0, 10, 0, 2.0, -1.0)
    y = pa.PVec(ctx, n).fill_hash(0, 11, 0, 2.0, -1.0)
    V = [pa.PVec(ctx, n).fill_hash(0, 20 + j, 0, 2.0, -1.0) for j in range(3)]
    first = (x.dot(y), x.maxabs(), tuple(x.mdot(V)))
    reps = 20000 if forced_rccl else 100000
    xh, yh = x.handle, y.handle
    out = C.c_double()
    for i in range(reps):
        if i % 1000 == 0:  # the slower Python paths now and then: max (all-gather form under RCCL) and a panel
            assert (x.dot(y), x.maxabs(), tuple(x.mdot(V))) == first
        else:
            check(lib.po_vec_dot(xh, yh, C.byref(out)))
            assert out.value == first[0]
    sc = ctx.sync_counters()
    assert sc["flag_waits"] >= reps and sc["flag_timeouts"] == 0, sc
    if forced_rccl:
        assert sc["allreduces"] >= reps - reps // 1000 and sc["allgathers"] >= reps // 1000, sc
    ctx.close()


def _worker_ckpt(rank, world, port, q, args, opts, path):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import paropt_amd as pa

    ctx = pa.Context(0)
    ctx.init_callback_from_torch()

    def make():
        p = pa.SeparableProblem(ctx, args["problem"], args["n"], args["c"])
        if args.get("chain_span", 0):
            p.setChain(args["chain_span"], args.get("chain_stride", 1))
        return p

    ip = pa.InteriorPoint(make(), opts)
    ip.optimize(checkpoint=path)  # the file left behind is the last multiple of write_output_frequency
    # a second file with the FINAL state, re-read on the same two ranks: bit-exact round trip of every block
    ip.writeSolutionFile(path + ".final")
    ip2 = pa.InteriorPoint(make(), opts)
    ip2.readSolutionFile(path + ".final")
    same = True
    for a, b in zip(ip.getOptimizedPoint(), ip2.getOptimizedPoint()):
        a = a.to_numpy() if hasattr(a, "to_numpy") else np.asarray(a)
        b = b.to_numpy() if hasattr(b, "to_numpy") else np.asarray(b)
        same = same and np.array_equal(a, b)
    if ip.getOptimizedSparse() is not None:
        for a, b in zip(ip.getOptimizedSparse()[:2], ip2.getOptimizedSparse()[:2]):
            same = same and np.array_equal(a.to_numpy(), b.to_numpy())
    flags = [None] * world
    dist.all_gather_object(flags, bool(same))
    if rank == 0:
        q.put(all(flags))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["ip_quadratic_checkpoint_n131_c3_r2", "ipcsr_convex_checkpoint_n121_c2_chain2_r2"])
def test_two_rank_solution_file_is_the_references_single_file(name, tmp_path):
    """writeSolutionFile on two ranks: ONE file in the layout of the concatenated problem, as the reference's
    MPI-IO code writes it (src/ParOptInteriorPoint.cpp:883-972) - compared with the file the reference left
    behind on two MPI ranks (header bit-exact, payload to 1e-6), re-read on two ranks (bit-exact), and, where
    the single-rank problem is the same problem, read by ONE rank."""
    import struct

    from conftest import ip_options_from_case, load_golden

    g, case = load_golden(name)
    ref = g["checkpoint_bytes"].tobytes()
    a = case["args"]
    opts = ip_options_from_case(case)
    path = str(tmp_path / "ckpt.bin")
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker_ckpt, args=(r, 2, port, q, a, opts, path)) for r in range(2)]
    for p in procs:
        p.start()
    roundtrip = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert roundtrip
    mine = open(path, "rb").read()
    assert not os.path.exists(path + ".0")
    assert len(mine) == len(ref) and mine[:12] == ref[:12]
    nv, nwc, c = struct.unpack("<3i", mine[:12])
    assert len(mine) == 12 + (5 * c + 1) * 8 + 3 * nv * 8 + 2 * nwc * 8
    pm = np.frombuffer(mine[12:], dtype="<f8")
    pr = np.frombuffer(ref[12:], dtype="<f8")
    np.testing.assert_allclose(pm, pr, rtol=1e-6, atol=1e-6 * np.abs(pr).max())
    if nwc == 0:
        import paropt_amd as pa

        ctx = pa.Context(0)
        ip = pa.InteriorPoint(pa.SeparableProblem(ctx, a["problem"], a["n"], a["c"]), opts)
        ip.readSolutionFile(path)
        np.testing.assert_array_equal(ip.getOptimizedPoint()[0].to_numpy(), pm[1 + 5 * c: 1 + 5 * c + nv])


def _worker_user(rank, world, port, q, deferred):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import paropt_amd as pa

    ctx = pa.Context(0)
    ctx.init_callback_from_torch()
    # po_ctx_allreduce: the MPI_Allreduce of user code, through the context's communicator
    a = ctx.allreduce(np.array([rank + 1.0, 10.0 * (rank + 1)]), "sum")
    b = ctx.allreduce(np.array([rank + 1.0]), "min")
    c = ctx.allreduce(np.array([rank + 1.0]), "max")
    assert list(a) == [world * (world + 1) / 2, 10.0 * world * (world + 1) / 2] and b[0] == 1.0 and c[0] == world
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                       "librandom_convex_user.so")
    user = pa.UserLibraryProblem(ctx, lib, 30011, 5)
    if deferred:
        user.setDeferredReductions(True)
    ip = pa.InteriorPoint(user, {"qn_type": "bfgs", "qn_subspace_size": 6, "abs_res_tol": 1e-8,
                                 "start_affine_multiplier_min": 0.01, "max_major_iters": 15,
                                 "write_output_frequency": 0})
    ip.optimize()
    x = ip.getOptimizedPoint()[0].to_numpy()
    xs = [None] * world
    dist.all_gather_object(xs, (user.offset, x))
    if rank == 0:
        q.put((ip.getIterationCounters(), ip.getObjective()[0],
               np.concatenate([v for _, v in sorted(xs, key=lambda t: t[0])])))
    dist.barrier()
    user.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("deferred", [False, True])
def test_user_library_problem_on_two_ranks(deferred):
    """The user-side problem of examples/random_convex_amd.cpp sharded over two ranks (its rank-local objective parts
    summed by po_ctx_allreduce, or -- deferred -- by po_ctx_reduce_device inside the solver's batch) against the
    same problem on one rank."""
    import paropt_amd as pa

    ctx = pa.Context(0)
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "examples",
                       "librandom_convex_user.so")
    user = pa.UserLibraryProblem(ctx, lib, 30011, 5)
    ip = pa.InteriorPoint(user, {"qn_type": "bfgs", "qn_subspace_size": 6, "abs_res_tol": 1e-8,
                                 "start_affine_multiplier_min": 0.01, "max_major_iters": 15,
                                 "write_output_frequency": 0})
    ip.optimize()
    ref = (ip.getIterationCounters(), ip.getObjective()[0], ip.getOptimizedPoint()[0].to_numpy())
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker_user, args=(r, 2, port, q, deferred)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert got[0] == ref[0]
    assert abs(got[1] - ref[1]) <= 1e-9 * max(1.0, abs(ref[1]))
    np.testing.assert_allclose(got[2], ref[2], rtol=0, atol=1e-7)
    user.close()


# ---- drawn cases (tests/test_gpu_random_sweep.py) sharded over three ranks ---------------------------------------------
def _drawn_dense_cases(limit):
    import test_gpu_random_sweep as T

    out = []
    for problem, n, c, opts, wt, extra in T.cases():
        if wt is None and not extra.get("chain") and n >= 3 and problem != "rosenbrock":
            out.append((problem, n, c, opts, extra))
        if len(out) == limit:
            break
    # ... and one with fewer variables than ranks: the third rank owns an EMPTY shard
    out.append(("quadratic", 2, 1, {"qn_subspace_size": 2, "qn_type": "bfgs", "abs_res_tol": 1e-8,
                                    "start_affine_multiplier_min": 0.01, "max_major_iters": 6}, {}))
    return out


def _run_drawn(pa, ctx, case):
    problem, n, c, opts, extra = case
    prob = pa.SeparableProblem(ctx, problem, n, c, extra.get("seed", 0), 1.0, extra.get("eig_max", 100.0))
    if extra.get("bounds_mode", 0):
        prob.setBoundsMode(extra["bounds_mode"])
    ip = pa.InteriorPoint(prob, dict(opts, write_output_frequency=0, max_major_iters=6))
    snaps = []
    ip.setIterationCallback(lambda k: snaps.append(ip.snapshot()))
    ip.optimize()
    return prob, [(tuple(int(v) for v in s["counters"]), int(s["qn_size"]), float(s["fobj"]), float(s["mu"]),
                   tuple(float(v) for v in s["norms"])) for s in snaps], ip.getOptimizedPoint()[0].to_numpy()


def _worker_drawn(rank, world, port, q, ncases):
    import torch.distributed as dist

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import paropt_amd as pa

    ctx = pa.Context(0)
    ctx.init_callback_from_torch()
    res = []
    for case in _drawn_dense_cases(ncases):
        prob, snaps, x = _run_drawn(pa, ctx, case)
        xs = [None] * world
        dist.all_gather_object(xs, (prob.offset, x))
        res.append((snaps, np.concatenate([a for _, a in sorted(xs, key=lambda t: t[0])])))
    if rank == 0:
        q.put(res)
    dist.barrier()
    dist.destroy_process_group()


def test_drawn_cases_on_three_ranks_match_single_rank():
    """Ten drawn dense cases (sizes around the tile sizes, all barrier strategies, norms, line-search and quasi-Newton
    switches) with the design vector sharded over THREE ranks -- uneven shards, odd shard lengths -- against the
    single-rank run: counters exactly, objective / barrier parameter / norms / the point to 1e-7."""
    import paropt_amd as pa

    ncases = 10
    ctx = pa.Context(0)
    single = [_run_drawn(pa, ctx, case)[1:] for case in _drawn_dense_cases(ncases)]
    mpctx = mp.get_context("spawn")
    q = mpctx.Queue()
    port = _free_port()
    procs = [mpctx.Process(target=_worker_drawn, args=(r, 3, port, q, ncases)) for r in range(3)]
    for p in procs:
        p.start()
    multi = q.get(timeout=600)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for case, (s1, x1), (s3, x3) in zip(_drawn_dense_cases(ncases), single, multi):
        assert len(s1) == len(s3), case
        for a, b in zip(s3, s1):
            assert a[0] == b[0] and a[1] == b[1], (case, a, b)
            assert abs(a[2] - b[2]) <= 1e-7 * max(1.0, abs(b[2])), (case, a, b)
            assert abs(a[3] - b[3]) <= 1e-7 * abs(b[3]), (case, a, b)
            np.testing.assert_allclose(a[4], b[4], rtol=1e-6, atol=1e-10, err_msg=repr(case))
        np.testing.assert_allclose(x3, x1, rtol=0, atol=1e-7, err_msg=repr(case))
