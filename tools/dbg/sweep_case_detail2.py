"""Per-iteration device / oracle values of one drawn case of tests/test_gpu_random_sweep.py.
usage: PAROPT_SWEEP_CASES=600 PAROPT_SWEEP_SEED=808 python tools/dbg/sweep_case_detail2.py 160"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_random_sweep as T  # noqa: E402
import paropt_amd as pa  # noqa: E402
from oracle import paropt_oracle as po  # noqa: E402

idx = int(sys.argv[1])
problem, n, c, opts, wt, extra = T.cases()[idx]
print(T.cases()[idx])
wargs = dict(nwcon=wt[0], nw=wt[1], nwstart=wt[2], nwskip=wt[3], nwineq=wt[4]) if wt else {}
wargs.update(extra)
bopt = wargs.pop("bound_options", None)
oprob = po.SepProblem(problem, n, c, **wargs)
if bopt:
    oprob.use_lower, oprob.use_upper = bool(bopt[0]), bool(bopt[1])
oip = po.InteriorPoint(oprob, opts)
osn = []
oip.hook = lambda s, k: osn.append(s.snapshot())
oip.optimize()
ctx = T._make_ctx()
prob = pa.SeparableProblem(ctx, problem, n, c, extra.get("seed", 0), 1.0, extra.get("eig_max", 100.0))
if wt:
    prob.setWeighting(*wt)
if extra.get("bounds_mode", 0):
    prob.setBoundsMode(extra["bounds_mode"])
if extra.get("chain"):
    prob.setChain(*extra["chain"])
if bopt:
    prob.setVarBoundOptions(*bopt)
ip = pa.InteriorPoint(prob, dict(opts, write_output_frequency=0))
gsn = []
ip.setIterationCallback(lambda k: gsn.append(ip.snapshot()))
ip.optimize()
toks = T.info_tokens(ip.getHistory())
for k in range(min(len(osn), len(gsn))):
    print("it %d device counters %s qn %s mu %.17g fobj %.17g norms %s z %s tokens %s" % (
        k, list(gsn[k]["counters"]), gsn[k]["qn_size"], gsn[k]["mu"], gsn[k]["fobj"], gsn[k]["norms"], gsn[k]["z"][:4], toks.get(k, [])))
    print("     oracle counters %s qn %s mu %.17g fobj %.17g norms %s z %s tokens %s" % (
        list(osn[k]["counters"]), osn[k]["qn_size"], osn[k]["mu"], osn[k]["fobj"], osn[k]["norms"], osn[k]["z"][:4],
        oip.trace[k]["info"].split() if k < len(oip.trace) else None))
