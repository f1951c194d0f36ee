"""Prints device and oracle snapshots of one draw of the random sweep side by side.
usage: PAROPT_SWEEP_CASES=N PAROPT_SWEEP_SEED=S python tools/dbg/sweep_case_detail.py i"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(R, "tests")); sys.path.insert(0, R)
import test_gpu_random_sweep as S
import paropt_amd as pa
from oracle import paropt_oracle as po
from test_gpu_ip import info_tokens
ctx = S._make_ctx()
idx = int(sys.argv[1])
problem, n, c, opts, wt, extra = S.cases()[idx]
print(problem, n, c, opts, wt, extra)
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
tok = info_tokens(ip.getHistory())
for k in range(min(len(osn), len(gsn), 8)):
    print("it %d device counters %s qn %s mu %.17g fobj %.17g norms %s tokens %s" % (k, list(gsn[k]["counters"]), gsn[k]["qn_size"], gsn[k]["mu"], gsn[k]["fobj"], np.array(gsn[k]["norms"]), tok.get(k)))
    print("     oracle counters %s qn %s mu %.17g fobj %.17g norms %s tokens %s" % (list(osn[k]["counters"]), osn[k]["qn_size"], osn[k]["mu"], osn[k]["fobj"], np.array(osn[k]["norms"]), oip.trace[k]["info"].split() if k < len(oip.trace) else None))
