import sys, time
sys.path.insert(0, '/root/repo')
import paropt_amd as pa
ctx = pa.Context(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 5000000
prob = pa.SeparableProblem(ctx, "quadratic", n, 4, 0)
tr = pa.TrustRegion(prob, {"qn_subspace_size": 10, "tr_max_iterations": 10})
tr.setEigenModelSynthetic(10, 0, 0, 2.0)
def cb(i):
    if i > 0:
        s = tr.getState(); print(i-1, s["subproblem_iters"], s["adaptive_subproblem_iters"], tr.getLastRow()[1], flush=True)
tr.setIterationCallback(cb)
tr.optimize()
s = tr.getState(); print("last", s["subproblem_iters"], s["adaptive_subproblem_iters"], tr.getLastRow()[1])
