import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import paropt_amd as pa

def run(env, problem="convex", qn="sr1", linear=True):
    for k in ("PAROPT_AMD_NO_FUSED_UPDATE", "PAROPT_AMD_NO_RECOMPUTE"):
        os.environ.pop(k, None)
    for k in env:
        os.environ[k] = "1"
    c = pa.Context(0)
    prob = pa.SeparableProblem(c, problem, 20011, 7)
    prob.setLinearConstraints(linear)
    ip = pa.InteriorPoint(prob, {"qn_type": qn, "qn_subspace_size": 6, "abs_res_tol": 1e-8,
                                 "start_affine_multiplier_min": 0.01, "max_major_iters": 8, "write_output_frequency": 0})
    out = []
    def cb(k):
        x, z, zl, zu = ip.getOptimizedPoint()
        out.append((ip.snapshot(), x.to_numpy().copy(), zl.to_numpy().copy(), zu.to_numpy().copy()))
    ip.setIterationCallback(cb)
    ip.optimize()
    return out

a = run([]); b = run(["PAROPT_AMD_NO_FUSED_UPDATE"])
for k, (sa, sb) in enumerate(zip(a, b)):
    print(k, "x", np.abs(sa[1]-sb[1]).max(), "zl", np.abs(sa[2]-sb[2]).max(), "zu", np.abs(sa[3]-sb[3]).max(),
          "norms", np.array(sa[0]["norms"]) - np.array(sb[0]["norms"]), "z", np.abs(np.array(sa[0]["z"])-np.array(sb[0]["z"])).max())
