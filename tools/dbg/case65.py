import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import paropt_amd as pa
ctx = pa.Context(0)
base = {'qn_subspace_size': 2, 'qn_type': 'sr1', 'abs_res_tol': 1e-08, 'start_affine_multiplier_min': 0.01, 'max_major_iters': 3, 'norm_type': 'l2',
        'sequential_linear_method': True, 'use_diag_hessian': True, 'qn_sigma': 1.0, 'starting_point_strategy': 'affine_step', 'penalty_gamma': 1000.0,
        'write_output_frequency': 0}
def run(tag, drop=(), n=129):
    opts = {k: v for k, v in base.items() if k not in drop}
    prob = pa.SeparableProblem(ctx, "quadratic", n, 17)
    prob.setWeighting(20, 2, 5, 1, 20)
    ip = pa.InteriorPoint(prob, opts)
    sn = []
    ip.setIterationCallback(lambda k: sn.append(ip.snapshot()))
    ip.optimize()
    print(tag, [round(float(s["fobj"]), 3) for s in sn], flush=True)
run("all (reference: 7640.885 7849.042 6165.217)")
run("no seq_lin", ("sequential_linear_method",))
run("no diag_hessian", ("use_diag_hessian",))
run("no qn_sigma", ("qn_sigma",))
run("n=130", (), 130)
