#!/usr/bin/env python3
"""One drawn case of the trust-region reference fixture (tests/test_gpu_tr_sweep.py) with given debug switches:
    python tools/dbg/tr_fixture_case.py 64 "14=0,15=0" "14=1,15=0" "14=1,15=1"
prints the device's table tokens beside the compiled reference's."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np  # noqa: E402

import paropt_amd as pa  # noqa: E402
from paropt_amd.lib import lib  # noqa: E402
import test_gpu_tr_sweep as T  # noqa: E402

idx = int(sys.argv[1])
g, drawn = T._fixture()
case = drawn[idx]
print(case)
ctx = pa.Context(0)
ref = json.loads(str(g["d%04d/tokens" % idx]))
print("reference:", ref, float(g["d%04d/fk" % idx][0]))
for var in sys.argv[2:] or [""]:
    sw = [(int(p.split("=")[0]), int(p.split("=")[1])) for p in var.split(",") if p]
    for i, v in sw:
        lib.po_debug_set_switch(i, v)
    problem, n, c, m, tro, wt, extra = case
    prob = pa.SeparableProblem(ctx, problem, n, c, extra.get("seed", 0), 1.0, extra.get("eig_max", 100.0))
    if wt:
        prob.setWeighting(*wt)
    if extra.get("chain"):
        prob.setChain(*extra["chain"])
    tr = pa.TrustRegion(prob, dict(tro, qn_subspace_size=m, max_major_iters=200, output_file="", tr_output_file=""))
    rows = []
    tr.setIterationCallback(lambda i: rows.append(tr.getLastRow()) if i > 0 else None)
    tr.optimize()
    rows.append(tr.getLastRow())
    print("switches %-14s:" % var, [t for _, t in rows], tr.getState()["fk"])
    for r, _ in rows:
        print("    ", r)
    for i, v in sw:
        lib.po_debug_set_switch(i, -1)
