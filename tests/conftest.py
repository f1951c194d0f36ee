import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu)")


def load_golden(name):
    """A fixture produced by oracle/make_golden.py from the compiled reference."""
    d = dict(np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False))
    case = json.loads(str(d.pop("case_json")))
    return d, case


def golden_names(prefix):
    return sorted(f[:-4] for f in os.listdir(GOLDEN_DIR) if f.startswith(prefix) and f.endswith(".npz"))


def ip_options_from_case(case):
    """Translate the ref_driver 'opt.*' arguments into an options dict."""
    opts = {}
    for k, v in case["args"].items():
        if k.startswith("opt."):
            opts[k[4:]] = v
    return opts


def golden_vec_view(full, case):
    """The part of a GLOBAL design-sized vector a golden holds: rank 0's shard of the reference run
    (contiguous blocks, oracle/ref_driver.cpp shard()), every vec_stride-th element."""
    n = int(case["args"]["n"])
    ranks = int(case.get("ranks", 1))
    nloc0 = n // ranks + (1 if n % ranks > 0 else 0)
    return np.asarray(full)[:nloc0:int(case["args"].get("vec_stride", 1))]


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN_DIR


# How far each short-window golden is compared.  The reference's own later iterations are driven by round-off
# level quantities in these runs (e.g. a penalty parameter rho = numer / (0.7 * infeas) with infeas ~ 1e-14 after an
# inexact Newton step; L-SR1 inside the line-search interior point does not converge, SURVEY.md 8d: the steps
# shrink to 1e-3..1e-6 and the state becomes a round-off artefact), so agreement cannot last for the whole record.
# tools/agreement_windows.py measures, on the GPU, the first iteration at which a check of
# tests/test_gpu_ip.py::test_ip_trajectory_golden fails when the window is the whole recorded run; the windows
# below are that measurement minus a margin of two iterations ("agrees_through" in the comments, recorded length in
# brackets; integers = counters, quasi-Newton size, pivots, clamp counts, info tokens).
GOLDEN_WINDOWS = {
    # device "agrees through" (tools/agreement_windows.py) | the REFERENCE agrees with ITSELF through
    # (oracle/reference_self_agreement.py: 1-4 MPI ranks, four BLAS code paths) | [recorded iterations]
    "ip_convex_hvec_n300_c3": 22,              # 24 | 24 [53]
    "ip_convex_hvec_noprecon_n200_c2": 14,     # 16 | 16 [80]
    "ipw_convex_n240_c3_w40_mpc": 16,          # state 18 (round 4: 20), integers 40 | 17 [60]
    # L-SR1 (the quasi-Newton type of the metric's configuration)
    "ip_convex_n300_c5_sr1": 14,               # 16 | 16 [25]
    "ip_convex_sigma_sr1_n300_c3": 18,         # 20 | 20 [25]
    "ip_convex_n2000_c32_sr1": 20,             # the whole record | the whole record [20]
    "ip_convex_n2000_c90_sr1": 20,             # the whole record | the whole record [20] (100-column panel)
    "ip_convex_n100000_c32_sr1_r4": 15,        # state 17 (round 4: 16), integers the whole record | 16 [20]; n = 1e5 on 4 MPI ranks
    "ipw_convex_n240_c3_w40_sr1": 22,          # 24 | 24 [60]
    "ipcsr_convex_n200_c2_chain5s3_sr1": 34,   # state 36 (round 4: 37), integers 46 (45) | 33 [60]
}


def reference_self_agreement():
    """{golden: first iteration at which the REFERENCE differs from ITSELF} -- the unmodified reference run on 1-4 MPI
    ranks and with four BLAS code paths (oracle/reference_self_agreement.py, measured in the build container,
    committed as profiles/r04_reference_self_agreement.jsonl).  Past that iteration the golden is one summation
    order's artefact."""
    import json

    out = {}
    path = os.path.join(ROOT, "profiles", "r04_reference_self_agreement.jsonl")
    with open(path) as f:
        for ln in f:
            if ln.strip():
                d = json.loads(ln)
                out[d["golden"]] = d["reference_self_agrees_through"]
    return out


# A short window is only legitimate where the reference itself is unstable: every hand-set window must reach to
# within two iterations of the point where the reference stops agreeing with itself, so that a product regression
# cannot hide inside a window that was merely measured on the product.
_SELF = reference_self_agreement()
for _name, _w in GOLDEN_WINDOWS.items():
    assert _name in _SELF and _SELF[_name] is not None, "no reference self-agreement record for %s" % _name
    assert _w >= _SELF[_name] - 2, "window of %s (%d) ends before the reference's own instability (%d)" % (
        _name, _w, _SELF[_name])


# The numpy oracle (oracle/paropt_oracle.py) is compared at a tighter tolerance (1e-7) and, being another
# implementation with another summation order, leaves the reference's round-off level trajectories at its own
# iteration: its windows are the hand-set ones of round 2 (L-SR1 goldens without an entry: 8).
GOLDEN_WINDOWS_ORACLE = {
    "ip_convex_hvec_n300_c3": 23,
    "ip_convex_hvec_noprecon_n200_c2": 15,
    "ipw_convex_n240_c3_w40_mpc": 15,
}


# ---- tolerance schedule of the trajectory comparisons (round 5) -----------------------------------------------------
# tests/golden/self_disagreement.json (oracle/reference_self_disagreement.py, build container): for every trajectory
# golden and every iteration k, by how much the UNMODIFIED REFERENCE differs from that golden when only the summation
# order of its reductions changes (three other BLAS code paths, 1-4 MPI ranks), in the normalisations of the checks.
# The device path is held to
#       tol(k) = max(1e-12, FACTOR x the largest self-disagreement of iterations 0..k),       FACTOR = 100
# -- not asked to follow a golden more closely than the reference follows itself, and not allowed the flat 1e-6 / 1e-5
# of rounds 1-4, eight orders above the agreement actually reached.  Measured on MI355X with
# tools/trajectory_errors.py (profiles/r05_trajectory_errors.json): the device's deviation is at most 0.41 x tol(k) on
# every golden and every quantity, except the two below, which get a larger factor with the reason stated.
SCHEDULE_FLOOR = 1e-12
SCHEDULE_FACTOR = 100.0
SCHEDULE_FACTOR_BY_GOLDEN = {
    # deliberately broken bounds: x is moved off its bounds by the repair, the first iterations are ill-conditioned;
    # device deviation 140 x the reference's self-disagreement at iteration 2 (6.4e-12 vs 4.6e-14), same ratio later
    "ip_convex_badbounds7_n300_c3": 300.0,
    # CSR constraints: the reference runs (golden and variants alike) use the driver's dense LAPACK factorization of
    # S = C + Aw D^-1 Aw^T (METIS absent: ParOptSparseCholesky cannot be built, VERDICT r3 / r4 "partial"), the device a
    # level-scheduled sparse Cholesky in elimination order -- another algorithm, not another summation order; from
    # iteration 34 on (predictor-corrector steps at mu ~ 1e-9) its deviation is 20 x tol(k) of the plain factor
    "ipcsr_convex_n200_c2_chain5s3_mpc": 3000.0,
}
_SD = None


def tolerance_schedule(name):
    """tol(quantity, k) for golden `name`; quantity in mu / fobj / norms / dense / wnorms / vec."""
    import json

    global _SD
    if _SD is None:
        with open(os.path.join(GOLDEN_DIR, "self_disagreement.json")) as f:
            _SD = json.load(f)
    assert name in _SD, "no reference self-disagreement record for %s (oracle/reference_self_disagreement.py)" % name
    d = _SD[name]
    factor = SCHEDULE_FACTOR_BY_GOLDEN.get(name, SCHEDULE_FACTOR)
    vec = sorted((int(k), v) for k, v in d["vec"].items())

    def tol(quantity, k):
        if quantity == "vec":
            worst = max([v for kk, v in vec if kk <= k] or [0.0])
        else:
            series = d[quantity]
            worst = max(series[: k + 1]) if series else 0.0  # (past the shortest variant: everything recorded)
        return max(SCHEDULE_FLOOR, factor * worst)

    return tol


def oracle_window(name, default):
    return GOLDEN_WINDOWS_ORACLE.get(name, default)


def golden_window(name, default):
    return GOLDEN_WINDOWS.get(name, default)
