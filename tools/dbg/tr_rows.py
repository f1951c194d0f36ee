import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import paropt_amd as pa
from conftest import golden_names, load_golden
from tr_helpers import parse_tr_table
import test_gpu_tr as T
ctx = pa.Context(0)
for name in golden_names("tr_"):
    g, case = load_golden(name)
    tr, rows, snaps, final = T.run_gpu_tr(ctx, case)
    ref = parse_tr_table(g["paropt_tr"])
    n = min(len(ref), len(rows))
    mism = [(k, " ".join(rows[k][1]), " ".join(ref[k][1])) for k in range(n) if rows[k][1] != ref[k][1]]
    print(name, "rows", n, "ref rows", len(ref), "mine", len(rows), "mismatching", len(mism))
    for m in mism[:40]:
        print("    ", m)
