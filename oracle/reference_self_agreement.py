#!/usr/bin/env python3
"""
TEST INFRASTRUCTURE ONLY (build container: needs oracle/_ref/ref_driver, i.e. /root/reference).

How far does the REFERENCE agree with ITSELF on the goldens whose compared window is shorter than the recorded run
(tests/conftest.py::GOLDEN_WINDOWS)?  Each such case is run through the unmodified reference on 1, 2, 3 and 4 MPI
ranks -- same problem, same data (every array is a pure function of the global index), only the summation order of
the reductions differs -- and the per-iteration records of every pair of runs are compared with the very checks
tests/test_gpu_ip.py::test_ip_trajectory_golden applies to the device path: integers (counters, quasi-Newton size,
LU pivots, clamp counts, info tokens) exactly, state (mu, fobj, norms, dense blocks) to 1e-6 / 1e-5.  The first
iteration at which two reference runs differ is `reference_self_agrees_through`: past it the recorded trajectory is
an artefact of one summation order and no implementation can be expected to follow it.  Cases whose sparse
constraints are rank-local in the driver (weighting groups with a global inequality count, CSR chains) cannot be
sharded into the same problem; for those -- and for every other case as well -- the same experiment is made with the
BLAS code path instead (MKL_CBWR = COMPATIBLE / SSE4_2 / AVX2 / default: another summation order inside the ddot
behind ParOptVec::dot and mdot, src/ParOptVec.cpp:124-170, at the recorded rank count).

    python oracle/reference_self_agreement.py [name-substring]   ->  JSON lines (committed as
                                                                      profiles/r04_reference_self_agreement.jsonl)
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from make_golden import read_rec, run_driver  # noqa: E402


def info_tokens(text):
    toks = {}
    for ln in str(text).splitlines():
        parts = ln.split()
        if len(parts) >= 15 and parts[0].isdigit():
            toks[int(parts[0])] = parts[15:]
    return toks


def run_reference(case, ranks, cbwr=None):
    args = dict(case["args"])
    # MKL's conditional-numerical-reproducibility switch selects the code path (hence the summation order) of the
    # BLAS ddot behind ParOptVec::dot / mdot (src/ParOptVec.cpp:124-170) and of LAPACK: the same reference, the same
    # rank count, another rounding of every reduction
    if cbwr:
        os.environ["MKL_CBWR"] = cbwr
    else:
        os.environ.pop("MKL_CBWR", None)
    with tempfile.TemporaryDirectory() as td:
        args["out"] = os.path.join(td, "out.rec")
        args["text"] = os.path.join(td, "paropt.out")
        run_driver("ip", args, ranks)
        rec = read_rec(args["out"])
        with open(args["text"]) as f:
            lines = [ln.rstrip("\n") for ln in f]
        start = next((i for i, ln in enumerate(lines) if ln.startswith("iter ")), 0)
        rec["paropt_out"] = "\n".join(lines[start:])
    return rec


def first_difference(a, b):
    """First iteration at which two reference runs differ (integer checks / state checks of the device test)."""
    ta, tb = info_tokens(a["paropt_out"]), info_tokens(b["paropt_out"])
    na = 1 + max(int(k[2:5]) for k in a if k.startswith("it") and k.endswith("/mu"))
    nb = 1 + max(int(k[2:5]) for k in b if k.startswith("it") and k.endswith("/mu"))
    first_int = first_state = None
    for k in range(min(na, nb)):
        p = "it%03d/" % k
        ok_int = np.array_equal(a[p + "counters"], b[p + "counters"])
        for key in ("qn_size", "gpiv", "mfpiv", "clamped"):
            if p + key in a and p + key in b:
                ok_int = ok_int and np.array_equal(a[p + key], b[p + key])
        if k >= 1:
            ok_int = ok_int and ta.get(k, []) == tb.get(k, [])
        rt = 1e-6
        ok_state = abs(a[p + "mu"][0] - b[p + "mu"][0]) <= rt * abs(b[p + "mu"][0])
        ok_state = ok_state and abs(a[p + "fobj"][0] - b[p + "fobj"][0]) <= rt * max(1.0, abs(b[p + "fobj"][0]))
        na_, nb_ = np.asarray(a[p + "norms"]), np.asarray(b[p + "norms"])
        used = ~(np.isnan(na_) | np.isnan(nb_))
        ok_state = ok_state and np.allclose(na_[used], nb_[used], rtol=rt, atol=0)
        for key in ("z", "s", "t", "zs", "zt"):
            ref = b[p + key]
            ok_state = ok_state and np.allclose(a[p + key], ref, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ref).max()))
        if first_int is None and not ok_int:
            first_int = k
        if first_state is None and not ok_state:
            first_state = k
    n = min(na, nb)
    return first_int, first_state, n


def shardable(case, ranks):
    a = case["args"]
    if a.get("chain_span", 0) > 0:
        return False  # the driver's chain constraints are rank-local: another problem on another rank count
    if a.get("nwcon", 0) > 0:
        return False  # rank-local groups with per-rank counts: the inequality split cannot be reproduced
    return a["n"] >= 8 * ranks


def main():
    from conftest import GOLDEN_WINDOWS, load_golden

    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    for name in sorted(GOLDEN_WINDOWS):
        if pat not in name:
            continue
        g, case = load_golden(name)
        rank_counts = [r for r in (1, 2, 3, 4) if shardable(case, r)]
        pairs, worst = [], None
        if len(rank_counts) >= 2:
            runs = {r: run_reference(case, r) for r in rank_counts}
            for i, r1 in enumerate(rank_counts):
                for r2 in rank_counts[i + 1:]:
                    fi, fs, n = first_difference(runs[r1], runs[r2])
                    agree = min(x for x in (fi, fs, n) if x is not None)
                    pairs.append({"ranks": [r1, r2], "first_integer_difference": fi, "first_state_difference": fs,
                                  "agree_through": agree, "iterations_compared": n})
                    worst = agree if worst is None else min(worst, agree)
        # the other perturbation, available for every case: the BLAS code path at the recorded rank count
        r0 = case.get("ranks", 1)
        paths = ["COMPATIBLE", "SSE4_2", "AVX2", None]
        pruns = {c: run_reference(case, r0, c) for c in paths}
        for i, c1 in enumerate(paths):
            for c2 in paths[i + 1:]:
                fi, fs, n = first_difference(pruns[c1], pruns[c2])
                agree = min(x for x in (fi, fs, n) if x is not None)
                pairs.append({"mkl_cbwr": [c1 or "default", c2 or "default"], "ranks": [r0, r0],
                              "first_integer_difference": fi, "first_state_difference": fs, "agree_through": agree,
                              "iterations_compared": n})
                worst = agree if worst is None else min(worst, agree)
        os.environ.pop("MKL_CBWR", None)
        print(json.dumps({"golden": name, "recorded_on_ranks": r0, "rank_counts": rank_counts,
                          "reference_self_agrees_through": worst, "pairs": pairs}), flush=True)


if __name__ == "__main__":
    main()
