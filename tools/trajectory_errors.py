#!/usr/bin/env python3
"""GPU box: how closely does the device path follow every interior-point golden, iteration by iteration?

Runs each golden of tests/test_gpu_ip.py::IP_CASES through the C ABI and writes, per iteration, the differences from
the golden in the normalisations of the test (and of oracle/reference_self_disagreement.py):
mu, fobj, norms, dense blocks, wnorms, vectors -> gpurun_out/trajectory_errors.json.  Compared offline with
tests/golden/self_disagreement.json to calibrate / audit the tolerance schedule of the trajectory test.

    python tools/trajectory_errors.py [name-substring]
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import paropt_amd as pa
    from conftest import golden_vec_view, load_golden
    from test_gpu_ip import IP_CASES, run_gpu

    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    ctx = pa.Context(0)
    out = {}
    for name in IP_CASES:
        if pat not in name:
            continue
        g, case = load_golden(name)
        ip, snaps = run_gpu(ctx, case, want_vectors=True)
        nref = 1 + max(int(k[2:5]) for k in g if k.startswith("it") and k.endswith("/mu"))
        n = min(nref, len(snaps))
        rec = {"mu": [], "fobj": [], "norms": [], "dense": [], "wnorms": [], "vec": {}, "int_agree_through": n}
        for k in range(n):
            p = "it%03d/" % k
            s = snaps[k]
            rec["mu"].append(abs(s["mu"] - g[p + "mu"][0]) / abs(g[p + "mu"][0]))
            rec["fobj"].append(abs(s["fobj"] - g[p + "fobj"][0]) / max(1.0, abs(g[p + "fobj"][0])))
            na, nb = np.asarray(s["norms"]), np.asarray(g[p + "norms"])
            used = ~(np.isnan(na) | np.isnan(nb)) & (nb != 0)
            rec["norms"].append(float((np.abs(na[used] - nb[used]) / np.abs(nb[used])).max()) if used.any() else 0.0)
            dmax = 0.0
            for key in ("z", "s", "t", "zs", "zt"):
                ref = g[p + key]
                if ref.size:
                    dmax = max(dmax, float(np.abs(s[key] - ref).max() / max(1.0, np.abs(ref).max())))
            rec["dense"].append(dmax)
            if p + "wnorms" in g:
                wa, wb = np.asarray(s["wnorms"]), np.asarray(g[p + "wnorms"])
                nz = wb != 0
                rec["wnorms"].append(float((np.abs(wa[nz] - wb[nz]) / np.abs(wb[nz])).max()) if nz.any() else 0.0)
            if p + "x" in g:
                vmax = 0.0
                keys = ("x", "zl", "zu") + (("zw", "sw", "tw", "zsw", "ztw") if p + "zw" in g else ())
                for key in keys:
                    if s.get(key) is None:
                        continue
                    ref = g[p + key]
                    mine = golden_vec_view(s[key], case) if key in ("x", "zl", "zu") else s[key]
                    vmax = max(vmax, float(np.abs(mine - ref).max() / max(1.0, np.abs(ref).max())))
                rec["vec"][str(k)] = vmax
            ok = np.array_equal(s["counters"], g[p + "counters"])
            for key in ("gpiv", "mfpiv", "clamped"):
                if p + key in g:
                    ok = ok and np.array_equal(np.asarray(s[key]), g[p + key])
            if not ok and rec["int_agree_through"] == n:
                rec["int_agree_through"] = k
        rec["n"] = n
        out[name] = rec
        print("%-44s n=%3d ints through %3d  max(first 8) mu %.1e dense %.1e norms %.1e" % (
            name, n, rec["int_agree_through"], max(rec["mu"][:8]), max(rec["dense"][:8]), max(rec["norms"][:8])),
            flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "trajectory_errors.json"), "w") as f:
        json.dump(out, f, sort_keys=True)


if __name__ == "__main__":
    main()
