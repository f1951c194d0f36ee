#!/usr/bin/env python3
"""TEST INFRASTRUCTURE (not part of the product): the numpy oracle against the COMPILED REFERENCE on the drawn cases of
tests/test_gpu_random_sweep.py -- the GPU tests compare the device path with the oracle on those draws, this closes
the triangle.  Runs only where /root/reference was compiled (oracle/_ref/ref_driver), i.e. in the build container.

    python oracle/fuzz_vs_reference.py [ncases] [seed]
"""
import os
import subprocess
import sys
import tempfile
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    seed = sys.argv[2] if len(sys.argv) > 2 else "20261003"
    os.environ["PAROPT_SWEEP_CASES"] = str(ncases)
    os.environ["PAROPT_SWEEP_SEED"] = seed
    stub = types.ModuleType("test_gpu_ip")  # (the sweep module imports a helper of the GPU tests)
    stub.info_tokens = lambda t: {}
    sys.modules["test_gpu_ip"] = stub
    import test_gpu_random_sweep as T
    from oracle import paropt_oracle as po
    from oracle.make_golden import DRIVER, read_rec

    env = dict(os.environ, MKL_NUM_THREADS="1", PATH="/opt/conda/bin:" + os.environ.get("PATH", ""))
    nbad = nskip = 0
    only = [int(v) for v in os.environ.get("FUZZ_ONLY", "").split(",") if v]  # FUZZ_ONLY=i,j: these draws only
    drawn = T.large_cases() if os.environ.get("FUZZ_LARGE") else T.cases()  # FUZZ_LARGE=1: the large-n draws
    for i, (problem, n, c, opts, wt, extra) in enumerate(drawn):
        if only and i not in only:
            continue
        wargs = dict(nwcon=wt[0], nw=wt[1], nwstart=wt[2], nwskip=wt[3], nwineq=wt[4]) if wt else {}
        wargs.update(extra)
        bopt = wargs.pop("bound_options", None)
        oprob = po.SepProblem(problem, n, c, **wargs)
        if bopt:
            oprob.use_lower, oprob.use_upper = bool(bopt[0]), bool(bopt[1])
        oip = po.InteriorPoint(oprob, opts)
        osn = []
        oip.hook = lambda s, k: osn.append(s.snapshot())
        try:
            oip.optimize()
        except np.linalg.LinAlgError:
            nskip += 1
            continue
        rec = os.path.join(tempfile.gettempdir(), "fuzz_%d.rec" % os.getpid())
        args = ["problem=%s" % problem, "n=%d" % n, "c=%d" % c, "out=%s" % rec]
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
        r = subprocess.run([DRIVER, "ip"] + args, env=env, capture_output=True, text=True, cwd=tempfile.gettempdir())
        if r.returncode != 0 or not os.path.exists(rec):
            print("CASE %d: reference driver failed: %s" % (i, (r.stderr or r.stdout)[-300:]))
            nbad += 1
            continue
        g = read_rec(rec)
        os.remove(rec)
        ncmp = min(len(osn), 6 if (opts["qn_type"] == "sr1" or
                                   opts.get("barrier_strategy") == "mehrotra_predictor_corrector") else 8)
        msg = None
        for k in range(ncmp):
            p = "it%03d/" % k
            if p + "counters" not in g:
                msg = "reference stopped at iteration %d, oracle has %d" % (k, len(osn))
                break
            if k > 2 and float(np.max(osn[k]["norms"])) < 1e-7:
                break
            if list(g[p + "counters"]) != list(osn[k]["counters"]):
                msg = "counters @%d: reference %s oracle %s" % (k, list(g[p + "counters"]), list(osn[k]["counters"]))
                break
            if abs(g[p + "fobj"][0] - osn[k]["fobj"]) > 1e-6 * max(1.0, abs(g[p + "fobj"][0])):
                msg = "fobj @%d: reference %r oracle %r" % (k, g[p + "fobj"][0], osn[k]["fobj"])
                break
            if abs(g[p + "mu"][0] - osn[k]["mu"]) > 1e-6 * abs(g[p + "mu"][0]):
                msg = "mu @%d: reference %r oracle %r" % (k, g[p + "mu"][0], osn[k]["mu"])
                break
            if not np.allclose(g[p + "norms"], osn[k]["norms"], rtol=1e-6, atol=1e-11, equal_nan=True):
                msg = "norms @%d: reference %s oracle %s" % (k, g[p + "norms"], osn[k]["norms"])
                break
        if msg:
            nbad += 1
            print("CASE %d %r\n     -> %s" % (i, (problem, n, c, opts, wt, extra), msg), flush=True)
    print("%d of %d drawn cases differ between the compiled reference and the oracle (%d skipped: the oracle's dense "
          "Cholesky gave up)" % (nbad, ncases, nskip))


if __name__ == "__main__":
    main()
