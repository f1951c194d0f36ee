# Config 5 in ONE gpurun call, twice each: the round-5 corrector sequence, the corrector solve in two launches + the affine
# complementarity from the polynomial, and + Dinv / t left behind by the residual pass (profiles/r06_ab_config5.jsonl).
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r06_ab_config5.jsonl
for v in "0 0 0" "1 1 0" "1 1 1" "0 0 0" "1 1 0" "1 1 1"; do
  set -- $v
  PAROPT_AMD_MPC_FUSE=$1 PAROPT_AMD_MPC_POLY=$2 PAROPT_AMD_SPEC_DT=$3 python tools/bench_tr.py --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
d['switches'] = {'PAROPT_AMD_MPC_FUSE': $1, 'PAROPT_AMD_MPC_POLY': $2, 'PAROPT_AMD_SPEC_DT': $3}
print(json.dumps({k: d.get(k) for k in ('switches', 'value', 'inner_ip_iterations', 'inner_ip_iterations_per_s', 'ms_per_inner_iteration', 'launches_per_inner_iteration', 'host_syncs_per_inner_iteration', 'seconds_min', 'seconds_max')}))
" | tee -a gpurun_out/r06_ab_config5.jsonl
done
