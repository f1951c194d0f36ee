set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python3 tools/bench_tr.py --no-cpu-baseline > /dev/null 2>&1   # warm the box
for round in 1 2 3; do
  for lib in base new; do
    if [ $lib = base ]; then export PAROPT_AMD_LIB=$PWD/paropt_amd/libbase.so; else unset PAROPT_AMD_LIB; fi
    python3 tools/bench_tr.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib $round: %.2f TR it/s  %.0f inner it/s  (%d inner)' % (d['value'], d['inner_ip_iterations_per_s'], d['inner_ip_iterations']))"
  done
done
