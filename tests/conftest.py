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


# Trajectories whose later iterations are driven by round-off level quantities in the REFERENCE
# itself (e.g. a penalty parameter rho = numer / (0.7 * infeas) with infeas ~ 1e-14 after an inexact
# Newton step): compared over the stated number of leading iterations only.
GOLDEN_WINDOWS = {
    "ip_convex_hvec_n300_c3": 23,
    "ip_convex_hvec_noprecon_n200_c2": 15,
    "ipw_convex_n240_c3_w40_mpc": 15,
}


def golden_window(name, default):
    return GOLDEN_WINDOWS.get(name, default)
