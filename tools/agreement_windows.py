#!/usr/bin/env python3
"""How far do the device trajectories agree with the reference-run goldens?  For every golden whose compared window
is shorter than the recorded run (L-SR1 cases, hand-set windows in tests/conftest.py) this runs the product on the GPU
for the whole recorded length and prints the first iteration at which each class of check of
tests/test_gpu_ip.py::test_ip_trajectory_golden would fail: integers (counters, quasi-Newton size, pivots, clamp
counts, info tokens) and state (mu, fobj, norms, dense multipliers, at the test's tolerances).
    python tools/agreement_windows.py [name-substring]  ->  JSON lines"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import GOLDEN_WINDOWS, golden_names, load_golden  # noqa: E402
import test_gpu_ip as T  # noqa: E402


def first_failures(g, snaps, hist, nref):
    toks, mine = T.info_tokens(g["paropt_out"]), T.info_tokens(hist)
    first_int, first_state = None, None
    for k in range(min(nref, len(snaps))):
        p, s = "it%03d/" % k, snaps[k]
        ok_int = np.array_equal(s["counters"], g[p + "counters"])
        if p + "qn_size" in g:
            ok_int = ok_int and s.get("qn_size", 0) == int(g[p + "qn_size"][0])
        for key in ("gpiv", "mfpiv", "clamped"):
            if p + key in g:
                ok_int = ok_int and np.array_equal(np.asarray(s[key]), g[p + key])
        if k >= 1:
            ok_int = ok_int and mine.get(k, []) == toks.get(k, [])
        rt = 1e-6
        ok_state = abs(s["mu"] - g[p + "mu"][0]) <= rt * abs(g[p + "mu"][0])
        ok_state = ok_state and abs(s["fobj"] - g[p + "fobj"][0]) <= rt * max(1.0, abs(g[p + "fobj"][0]))
        used = ~np.isnan(s["norms"])
        ok_state = ok_state and np.allclose(s["norms"][used], g[p + "norms"][used], rtol=rt, atol=0)
        for key in ("z", "s", "t", "zs", "zt"):
            ref = g[p + key]
            ok_state = ok_state and np.allclose(s[key], ref, rtol=1e-5, atol=1e-5 * max(1.0, np.abs(ref).max()))
        if first_int is None and not ok_int:
            first_int = k
        if first_state is None and not ok_state:
            first_state = k
    return first_int, first_state


def main():
    import paropt_amd as pa

    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    ctx = pa.Context(0)
    names = [n for n in T.IP_CASES if ("sr1" in n or n in GOLDEN_WINDOWS) and pat in n]
    for name in names:
        g, case = load_golden(name)
        nref = 1 + max(int(k[2:5]) for k in g if k.startswith("it") and k.endswith("/mu"))
        ip, snaps = T.run_gpu(ctx, case, want_vectors=False)
        fi, fs = first_failures(g, snaps, ip.getHistory(), nref)
        print(json.dumps({"golden": name, "recorded_iterations": nref, "device_iterations": len(snaps),
                          "first_integer_mismatch": fi, "first_state_mismatch": fs,
                          "agrees_through": min(x for x in (fi, fs, min(nref, len(snaps))) if x is not None)}),
              flush=True)


if __name__ == "__main__":
    main()
