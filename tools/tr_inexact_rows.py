#!/usr/bin/env python3
"""Prints, for the trust-region goldens with allow-listed rows (tests/test_gpu_tr.py::TR_INEXACT_ROWS), how the two
interior-point solves of those rows ended on the device (last line of each solve's iteration table)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_golden  # noqa: E402
import test_gpu_tr as T  # noqa: E402
import paropt_amd as pa  # noqa: E402

ctx = pa.Context(0)
for name, rows in T.TR_INEXACT_ROWS.items():
    g, case = load_golden(name)
    lines = {}
    orig = T.run_gpu_tr
    import paropt_amd as pa2

    a = case["args"]
    tr, rws, snaps, final = T.run_gpu_tr(ctx, case, capture_lines=lines)
    for k in sorted(rows):
        print(json.dumps({"golden": name, "row": k, "info": rws[k][1], "steering": lines.get(k, ["", ""])[0],
                          "qp": lines.get(k, ["", ""])[1]}), flush=True)
