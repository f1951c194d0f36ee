set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for pc in 1 0 1 0; do
  PAROPT_AMD_WGRAM_PC=$pc python3 tools/bench_tr.py --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('WGRAM_PC=$pc: %.2f TR it/s  %.0f inner it/s  (%d inner)' % (d['value'], d['inner_ip_iterations_per_s'], d['inner_ip_iterations']))"
done
rm -rf gpurun_out/trace_c5
PAROPT_AMD_WGRAM_PC=0 rocprofv3 --kernel-trace --stats -d gpurun_out/trace_c5 -o t --output-format csv -- python3 tools/bench_tr.py --no-cpu-baseline --repeats 1 > /dev/null 2> gpurun_out/trace_c5.err
grep wgram gpurun_out/trace_c5/*kernel_stats.csv | cut -d, -f1-4 | sed 's/(double const.*)"//' | cut -c1-120
rm -rf gpurun_out/trace_c5
