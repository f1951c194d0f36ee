import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import paropt_amd as pa
ctx = pa.Context(0)
problem, qn, strategy = sys.argv[1:4]
opts = {"qn_type": qn, "qn_subspace_size": 6, "abs_res_tol": 1e-8, "start_affine_multiplier_min": 0.01,
        "max_major_iters": 45, "write_output_frequency": 0, "barrier_strategy": strategy}
runs = []
for flag in (False, True):
    prob = pa.SeparableProblem(ctx, problem, 20011, 7)
    prob.setLinearConstraints(flag)
    ip = pa.InteriorPoint(prob, opts)
    sn = []
    ip.setIterationCallback(lambda k: sn.append(ip.snapshot()))
    ip.optimize()
    runs.append((sn, ip.getHistory()))
a, b = runs
for k, (sa, sb) in enumerate(zip(a[0], b[0])):
    print(k, sa["counters"], sb["counters"], "%.3e" % abs(sa["fobj"] - sb["fobj"]), np.abs(np.array(sa["norms"]) - np.array(sb["norms"])), "%.3e"%np.abs(np.array(sa["z"]) - np.array(sb["z"])).max())
print(a[1][-1500:])
