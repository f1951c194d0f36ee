#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not part of the product): runs the drawn cases of tests/test_gpu_random_sweep.py through the
COMPILED REFERENCE (oracle/_ref/ref_driver: the unmodified sources of /root/reference) and stores what it did --
per-iteration evaluation counters, quasi-Newton size, barrier parameter, objective, the three norms, the dense
multipliers and the info tokens of its iteration table -- as a fixture: tests/golden/sweep_reference_s<seed>_n<N>.npz.
tests/test_gpu_random_sweep.py::test_random_case_against_reference_fixture then holds the DEVICE to the reference itself
on drawn cases (no numpy oracle in between).  Runs only where the reference was compiled (the build container).

    python oracle/make_sweep_reference.py [ncases=400] [seed=424242]
    python oracle/make_sweep_reference.py --large [ncases=40] [seed=434343]     (n = 32 769 ... 393 217)
"""
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def driver_args(problem, n, c, opts, wt, extra):
    args = ["problem=%s" % problem, "n=%d" % n, "c=%d" % c]
    if wt:
        args += ["nwcon=%d" % wt[0], "nw=%d" % wt[1], "nwstart=%d" % wt[2], "nwskip=%d" % wt[3], "nwineq=%d" % wt[4]]
    if "seed" in extra:
        args.append("seed=%d" % extra["seed"])
    if "eig_max" in extra:
        args.append("eig_max=%r" % extra["eig_max"])
    if "bounds_mode" in extra:
        args.append("bounds_mode=%d" % extra["bounds_mode"])
    if "bound_options" in extra:
        args += ["use_lower=%d" % extra["bound_options"][0], "use_upper=%d" % extra["bound_options"][1]]
    if "chain" in extra:
        args += ["chain_span=%d" % extra["chain"][0], "chain_stride=%d" % extra["chain"][1]]
    for k, v in opts.items():
        args.append("opt.%s=%s" % (k, int(v) if isinstance(v, bool) else v))
    args.append("opt.write_output_frequency=1")  # (the driver records an iteration where the reference writes output)
    return args


def main():
    argv = [a for a in sys.argv[1:] if a != "--large"]
    large = "--large" in sys.argv  # the draws at n = 32 769 ... 393 217 (large_cases_for): sweep_reference_large_s<seed>_n<N>.npz
    ncases = int(argv[0]) if len(argv) > 0 else (40 if large else 400)
    seed = int(argv[1]) if len(argv) > 1 else (434343 if large else 424242)
    import test_gpu_random_sweep as T  # (its info_tokens, from tests/test_gpu_ip.py, parses the reference's table too)
    from oracle.make_golden import DRIVER, read_rec

    env = dict(os.environ, MKL_NUM_THREADS="1", PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
    out = {}
    drawn = T.large_cases_for(seed, ncases) if large else T.cases_for(seed, ncases)
    nfail = 0
    for i, case in enumerate(drawn):
        problem, n, c, opts, wt, extra = case
        with tempfile.TemporaryDirectory() as td:
            rec, txt = os.path.join(td, "out.rec"), os.path.join(td, "paropt.out")
            r = subprocess.run([DRIVER, "ip"] + driver_args(*case) + ["out=" + rec, "text=" + txt], env=env,
                               capture_output=True, text=True, cwd=td)
            if r.returncode != 0 or not os.path.exists(rec):
                nfail += 1
                out["d%04d/failed" % i] = np.array([1])
                continue
            g = read_rec(rec)
            table = open(txt).read() if os.path.exists(txt) else ""
        K = 0
        while "it%03d/counters" % K in g:
            K += 1
        K = min(K, 12)
        pre = "d%04d/" % i
        out[pre + "counters"] = np.array([g["it%03d/counters" % k] for k in range(K)], dtype=np.int64).reshape(K, 3)
        out[pre + "qn_size"] = np.array([g["it%03d/qn_size" % k][0] for k in range(K)], dtype=np.int64)
        out[pre + "mu"] = np.array([g["it%03d/mu" % k][0] for k in range(K)])
        out[pre + "fobj"] = np.array([g["it%03d/fobj" % k][0] for k in range(K)])
        out[pre + "norms"] = np.array([g["it%03d/norms" % k] for k in range(K)]).reshape(K, 3)
        out[pre + "z"] = np.array([g["it%03d/z" % k] for k in range(K)]).reshape(K, -1)
        toks = T.info_tokens(table)
        out[pre + "tokens"] = np.array(json.dumps({int(k): v for k, v in toks.items() if int(k) < K}))
    out["cases_repr"] = np.array(json.dumps([repr(cs) for cs in drawn]))
    out["meta"] = np.array(json.dumps({"seed": seed, "ncases": ncases, "what": "compiled reference (oracle/_ref/ref_driver ip) on the "
                                       "draws of tests/test_gpu_random_sweep.py::cases_for(seed, ncases)", "driver_failures": nfail}))
    path = os.path.join(ROOT, "tests", "golden", "sweep_reference_%ss%d_n%d.npz" % ("large_" if large else "", seed, ncases))
    np.savez_compressed(path, **out)
    print("%s: %d draws, %d driver failures, %d bytes" % (path, ncases, nfail, os.path.getsize(path)))


if __name__ == "__main__":
    main()
