#!/usr/bin/env python3
"""A/B of kernel variants inside ONE process (cdna guide rule 24: interleaved rounds, one process, report the
distribution): the variants are settings of the library's debug switches (core.hpp DbgSwitch, po_debug_set_switch).

    python tools/ab_switch.py --variants "0=0;0=1" --rounds 4 --what micro --filter wgram
    python tools/ab_switch.py --variants "0=0,1=0;0=1,1=1" --rounds 3 --what iter

micro: po_bench_kernels (every hot kernel in isolation, n = 50 M, c = 32, k = 10); iter: the metric's interior-point
iteration (config 3, Jacobian rewritten at every gradient call), ms per iteration over 20 iterations after 12."""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", required=True, help='e.g. "0=0;0=1": switch id = value, comma separated, per variant')
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--what", default="micro", choices=["micro", "iter"])
    ap.add_argument("--filter", default="")
    ap.add_argument("--n", type=int, default=50_000_000)
    ap.add_argument("--c", type=int, default=32)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--qn", default="sr1")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--problem", default="convex")
    ap.add_argument("--nwcon", type=int, default=0, help="iter: sparse weighting constraints (config 4)")
    ap.add_argument("--nw", type=int, default=20)
    a = ap.parse_args()
    import paropt_amd as pa
    from paropt_amd.lib import lib

    variants = []
    for v in a.variants.split(";"):
        variants.append([(int(p.split("=")[0]), int(p.split("=")[1])) for p in v.split(",") if p])

    def apply(v):
        for i in range(17):
            lib.po_debug_set_switch(i, -1)
        for i, val in v:
            lib.po_debug_set_switch(i, val)

    ctx = pa.Context(0)
    res = {vi: {} for vi in range(len(variants))}
    if a.what == "micro":
        for r in range(a.rounds):
            for vi, v in enumerate(variants):
                apply(v)
                for row in pa.bench_kernels(ctx, a.n, a.c, a.k, a.reps):
                    if a.filter in row["kernel"]:
                        res[vi].setdefault(row["kernel"], []).append(row["avg_ms"])
    else:
        W, K = 12, 20
        prob = pa.SeparableProblem(ctx, a.problem, a.n, a.c, 0)
        if a.nwcon > 0:
            prob.setWeighting(a.nwcon, a.nw, 0, 0)
        prob.setLinearConstraints(False)
        opts = {"qn_type": a.qn, "qn_subspace_size": a.k, "abs_res_tol": 1e-30, "start_affine_multiplier_min": 0.01,
                "max_major_iters": W + K, "write_output_frequency": 0}
        ip = pa.InteriorPoint(prob, opts)
        st = {}

        def cb(k):
            if k == W:
                ctx.synchronize()
                ctx.time_wgram(True)
                st["t0"] = time.perf_counter()

        ip.setIterationCallback(cb)
        for r in range(a.rounds + 1):
            for vi, v in enumerate(variants):
                apply(v)
                ip.resetQuasiNewtonHessian()
                ip.optimize()
                ctx.synchronize()
                ms = 1e3 * (time.perf_counter() - st["t0"]) / K
                wg = [ctx.time_wgram_result(w) for w in (0, 1)]
                ctx.time_wgram(False)
                if r == 0:
                    continue  # warm-up round
                res[vi].setdefault("ms_per_iter", []).append(ms)
                ph = ip.getPhaseTimes()
                for kname, sec in ph.items():
                    res[vi].setdefault("phase:" + kname, []).append(1e3 * sec / (W + K))
                for w in (0, 1):
                    if wg[w][1] > 0:
                        res[vi].setdefault("wgram_launch_ms[%d]" % w, []).append(wg[w][0] / wg[w][1])
    for vi, v in enumerate(variants):
        for kname, xs in res[vi].items():
            print(json.dumps({"variant": a.variants.split(";")[vi], "what": kname, "median": statistics.median(xs),
                              "min": min(xs), "max": max(xs), "samples": [round(x, 4) for x in xs]}), flush=True)


if __name__ == "__main__":
    main()
