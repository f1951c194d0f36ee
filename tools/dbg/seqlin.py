import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import paropt_amd as pa
ctx = pa.Context(0)
prob = pa.SeparableProblem(ctx, "convex", 2049, 3)
opts = {'qn_subspace_size': 10, 'qn_type': 'bfgs', 'abs_res_tol': 1e-08, 'start_affine_multiplier_min': 0.01, 'max_major_iters': 4,
        'barrier_strategy': 'monotone', 'norm_type': 'infinity', 'sequential_linear_method': True, 'write_output_frequency': 0}
opts.update(eval(sys.argv[1]) if len(sys.argv) > 1 else {})
ip = pa.InteriorPoint(prob, opts)
sn = []
ip.setIterationCallback(lambda k: sn.append(ip.snapshot()))
ip.optimize()
print(os.environ.get("TAG", ""), [list(s["counters"]) for s in sn], [round(float(s["fobj"]), 4) for s in sn])
print(ip.getHistory()[-900:])
