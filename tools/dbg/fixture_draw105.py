"""Draw 105 of the compiled reference's fixture (tests/golden/sweep_reference_s424242_n400.npz), the one the suite skips by
name: a CSR chain whose diagonal block goes indefinite.  Prints the device's iterations and its iteration table (the
sparse Cholesky reports the breakdown).  usage: python tools/dbg/fixture_draw105.py"""
import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import test_gpu_random_sweep as T
import paropt_amd as pa
ctx = pa.Context(0)
case = T.cases_for(424242, 400)[105]
gsn, tok = T._run_device(ctx, case)
for k, s in enumerate(gsn[:6]):
    print(k, list(s["counters"]), s["qn_size"], s["mu"], s["fobj"], s["norms"], tok.get(k))
problem, n, c, opts, wt, extra = case
prob = pa.SeparableProblem(ctx, problem, n, c, extra.get("seed", 0), 1.0, 100.0)
prob.setChain(*extra["chain"])
ip = pa.InteriorPoint(prob, dict(opts, write_output_frequency=1))
ip.optimize()
print(ip.getHistory()[-2500:])
