#!/bin/bash
# Regenerates the rocprofv3 evidence under gpurun_out/ on the GPU box (copy what is to be kept into
# profiles/ afterwards).  One rocprofv3 run per configuration, kernel trace + stats only; the large
# per-dispatch traces are deleted so that only the summaries travel back.
#   usage (from the repo root on the GPU box):  bash tools/collect_profiles.sh
set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
run() {  # name, command...
  local name=$1
  shift
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$name -o $name --output-format csv -- "$@" \
      > gpurun_out/bench_under_rocprof_$name.json 2> gpurun_out/prof_$name.err
  rm -f gpurun_out/prof_$name/*_kernel_trace.csv
}
run c3 python3 bench.py --steps 20 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin
run c2 python3 bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin
run c4 python3 bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin
run c5 python3 tools/bench_tr.py --no-cpu-baseline
run csr python3 tools/bench_csr.py
ls -la gpurun_out/prof_c*/
