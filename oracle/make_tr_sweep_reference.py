#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not part of the product): runs the drawn cases of tests/test_gpu_tr_sweep.py through the COMPILED
REFERENCE (oracle/_ref/ref_driver tr: ParOptTrustRegion over ParOptQuadraticSubproblem, unmodified sources) and stores
its iteration count, final objective, final point norm and the info column of every row of its table as a fixture:
tests/golden/sweep_tr_reference_s<seed>_n<N>.npz.  tests/test_gpu_tr_sweep.py::
test_random_trust_region_case_against_reference_fixture then holds the DEVICE's driver to the reference itself.

    python oracle/make_tr_sweep_reference.py [ncases=150] [seed=535353]
"""
import importlib.util
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def driver_args(problem, n, c, m, tro, wt, extra):
    args = {"problem": problem, "n": n, "c": c, "seed": extra.get("seed", 0), "eig_max": extra.get("eig_max", 100.0),
            "opt.qn_subspace_size": m, "opt.max_major_iters": 200}
    if wt:
        args.update(nwcon=wt[0], nw=wt[1], nwstart=wt[2], nwskip=wt[3], nwineq=wt[4])
    if extra.get("chain"):
        args.update(chain_span=extra["chain"][0], chain_stride=extra["chain"][1])
    for k, v in tro.items():
        args[("opt." if k == "penalty_gamma" else "tr.") + k] = v
    return args


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 150
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 535353
    spec = importlib.util.spec_from_file_location("tr_sweep", os.path.join(ROOT, "tests", "test_gpu_tr_sweep.py"))
    T = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(T)
    from oracle.make_golden import read_rec, run_driver
    from tr_helpers import parse_tr_table

    drawn = T.cases_for(seed, ncases)
    out = {}
    for i, case in enumerate(drawn):
        args = driver_args(*case)
        with tempfile.TemporaryDirectory() as td:
            args["out"] = os.path.join(td, "out.rec")
            args["text"] = os.path.join(td, "paropt.tr")
            run_driver("tr", args)
            rec = read_rec(args["out"])
            table = parse_tr_table(open(args["text"]).read())
        pre = "d%04d/" % i
        out[pre + "iter_count"] = np.array([int(rec["final/iter_count"][0])])
        out[pre + "fk"] = np.array([float(rec["final/fk"][0])])
        out[pre + "xnorm"] = np.array([float(np.linalg.norm(rec["final/x"]))]) if "final/x" in rec else np.array([np.nan])
        out[pre + "tokens"] = np.array(json.dumps([table[k][1] for k in sorted(table)]))
        out[pre + "rows"] = np.array([table[k][0] for k in sorted(table)])
    out["cases_repr"] = np.array(json.dumps([repr(cs) for cs in drawn]))
    out["meta"] = np.array(json.dumps({"seed": seed, "ncases": ncases,
                                       "what": "compiled reference (oracle/_ref/ref_driver tr) on the draws of "
                                               "tests/test_gpu_tr_sweep.py::cases_for(seed, ncases)"}))
    path = os.path.join(ROOT, "tests", "golden", "sweep_tr_reference_s%d_n%d.npz" % (seed, ncases))
    np.savez_compressed(path, **out)
    print("%s: %d draws, %d bytes" % (path, ncases, os.path.getsize(path)))


if __name__ == "__main__":
    main()
