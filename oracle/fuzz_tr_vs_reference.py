#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not part of the product): the numpy trust-region restatement (oracle/tr_oracle.py) against the
COMPILED REFERENCE on the drawn cases of tests/test_gpu_tr_sweep.py -- the GPU test compares the device's driver with
the oracle on those draws, this closes the triangle (and adjudicates a differing draw: who differs from whom).  Runs
only where /root/reference was compiled (oracle/_ref/ref_driver), i.e. in the build container.

    python oracle/fuzz_tr_vs_reference.py [ncases] [seed]          FUZZ_ONLY=i,j: these draws only
"""
import importlib.util
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    if len(sys.argv) > 2:
        os.environ["PAROPT_TR_SWEEP_SEED"] = sys.argv[2]
    os.environ["PAROPT_TR_SWEEP_CASES"] = str(ncases)
    spec = importlib.util.spec_from_file_location("tr_sweep", os.path.join(ROOT, "tests", "test_gpu_tr_sweep.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    from oracle import paropt_oracle as po
    from oracle import tr_oracle as tro_mod
    from oracle.make_golden import read_rec, run_driver
    from tr_helpers import parse_tr_table

    only = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]
    nbad = nrun = 0
    for i, (problem, n, c, m, tro, wt, extra) in enumerate(T.cases()):
        if only and i not in only:
            continue
        okw = dict(extra)
        if wt:
            okw.update(nwcon=wt[0], nw=wt[1], nwstart=wt[2], nwskip=wt[3], nwineq=wt[4])
        ops = po.VecOps(po.SelfComm())
        sub = tro_mod.QuadraticSubproblem(po.SepProblem(problem, n, c, **okw),
                                          po.LBFGS(n, m, ops, "skip_negative_curvature"))
        otr = tro_mod.TrustRegion(sub, po.InteriorPoint(sub, T.oracle_ip_options(tro)), dict(tro))
        try:
            otr.optimize()
        except np.linalg.LinAlgError:
            continue
        args = {"problem": problem, "n": n, "c": c, "seed": extra.get("seed", 0), "eig_max": extra.get("eig_max", 100.0),
                "opt.qn_subspace_size": m, "opt.max_major_iters": 200}
        if wt:
            args.update(nwcon=wt[0], nw=wt[1], nwstart=wt[2], nwskip=wt[3], nwineq=wt[4])
        if extra.get("chain"):
            args.update(chain_span=extra["chain"][0], chain_stride=extra["chain"][1])
        for k, v in tro.items():
            args[("opt." if k == "penalty_gamma" else "tr.") + k] = v
        with tempfile.TemporaryDirectory() as td:
            args["out"] = os.path.join(td, "out.rec")
            args["text"] = os.path.join(td, "paropt.tr")
            run_driver("tr", args)
            rec = read_rec(args["out"])
            table = parse_tr_table(open(args["text"]).read())
        nrun += 1
        ref_tokens = [table[k][1] for k in sorted(table)]
        mine = [list(t["info"]) for t in otr.trace]
        ref_iters = int(rec["final/iter_count"][0])
        fk = float(rec["final/fk"][0])
        # (the reference prints one table row per iteration it started; the oracle's trace likewise)
        strip = lambda rows: [[t.split("/")[0] if "/" in t and t.replace("/", "").isdigit() else t for t in r] for r in rows]  # noqa: E731
        if not os.environ.get("FUZZ_TR_STEERING_COUNTS"):  # (default: the steering solve's own count is not compared,
            ref_tokens, mine = strip(ref_tokens), strip(mine)  #  see tests/test_gpu_tr_sweep.py)
        ok = ref_tokens == mine and ref_iters == otr.iter_count and abs(fk - sub.fk) <= 1e-6 * max(1.0, abs(fk))
        if not ok:
            nbad += 1
            print("TR CASE %d %r\n     -> reference iters %d fk %.12g tokens %s\n        oracle    iters %d fk %.12g tokens %s" % (
                i, (problem, n, c, m, tro, wt, extra), ref_iters, fk, ref_tokens, otr.iter_count, sub.fk, mine), flush=True)
    print("%d of %d drawn trust-region cases differ between the compiled reference and the oracle" % (nbad, nrun))


if __name__ == "__main__":
    main()
