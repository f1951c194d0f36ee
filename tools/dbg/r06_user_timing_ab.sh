cd $GRAFT_REPO_ROOT
for v in 1 0 1 0; do
  if [ $v = 1 ]; then export PAROPT_AMD_USER_TIMING=1; else unset PAROPT_AMD_USER_TIMING; fi
  python tools/bench_tr.py --no-cpu-baseline --repeats 3 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print('user_timing=$v', json.dumps({k: d.get(k) for k in ('value', 'inner_ip_iterations', 'inner_ip_iterations_per_s', 'ms_per_inner_iteration', 'launches_per_inner_iteration', 'host_syncs_per_inner_iteration', 'seconds_min', 'seconds_max')}))
"
done
for v in 1 0; do
  if [ $v = 1 ]; then export PAROPT_AMD_USER_TIMING=1; else unset PAROPT_AMD_USER_TIMING; fi
  python bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 --boundary builtin --no-cpu-baseline --repeats 3 --skip-extension-variant 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('c2 user_timing=$v', d['value'], d['ms_per_step'], d['user_eval_ms_per_iter'])"
  python bench.py --nglobal 6250000 --steps 20 --warmup 5 --boundary builtin --no-cpu-baseline --repeats 3 --skip-extension-variant 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1]); print('c3/8 user_timing=$v', d['value'], d['ms_per_step'], d['user_eval_ms_per_iter'])"
done
