#!/bin/bash
# Round-6 line of record of ONE configuration, all in one gpurun call (box-to-box spread does not enter a comparison):
#   1. the unprofiled bench line with the CPU reference beside it      -> gpurun_out/r06_bench_<cfg>.json
#   2. the same command under rocprofv3 --kernel-trace --stats         -> gpurun_out/r06_rocprofv3_kernel_stats_<cfg>.csv
#   3. PMC passes (FETCH_SIZE, WRITE_SIZE; counters only + kernel trace) -> gpurun_out/r06_pmc_<cfg>.json / .txt
#   usage (repo root on the GPU box):  bash tools/collect_r06.sh c2|c3|c3l|c4|c5 [nopmc]   (c3l: config 3 with L-BFGS(20), the 73-column panel)
set -u
cfg=${1:-c3}
pmc=${2:-pmc}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
case $cfg in
  c3) prog=bench.py; args="--steps 20 --warmup 5" ;;
  c3l) prog=bench.py; args="--qn bfgs --qn-size 20 --steps 20 --warmup 22 --boundary builtin" ;;
  c2) prog=bench.py; args="--nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 --boundary builtin" ;;
  c4) prog=bench.py; args="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 --boundary both" ;;
  c5) prog=tools/bench_tr.py; args="" ;;
  *) echo "unknown configuration $cfg"; exit 2 ;;
esac
python3 $prog $args > gpurun_out/r06_bench_$cfg.json 2> gpurun_out/r06_bench_$cfg.err
quick="$args --no-cpu-baseline"
if [ $prog = bench.py ]; then quick="$quick --repeats 1 --skip-extension-variant --boundary builtin"; else quick="$quick --repeats 1"; fi
rm -rf gpurun_out/prof_$cfg
rocprofv3 --kernel-trace --stats -d gpurun_out/prof_$cfg -o $cfg --output-format csv -- python3 $prog $quick \
    > gpurun_out/r06_bench_under_rocprof_$cfg.json 2> gpurun_out/prof_$cfg.err
cp gpurun_out/prof_$cfg/*kernel_stats.csv gpurun_out/r06_rocprofv3_kernel_stats_$cfg.csv
rm -rf gpurun_out/prof_$cfg
if [ $pmc = pmc ]; then
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rm -rf gpurun_out/pmc_${cfg}_$ctr
    rocprofv3 --pmc $ctr --kernel-trace -d gpurun_out/pmc_${cfg}_$ctr -o p --output-format csv -- python3 $prog $quick \
        > /dev/null 2> gpurun_out/pmc_${cfg}_$ctr.err
  done
  python3 tools/pmc_summary.py gpurun_out/r06_pmc_$cfg.json gpurun_out/pmc_${cfg}_FETCH_SIZE gpurun_out/pmc_${cfg}_WRITE_SIZE
  python3 tools/pmc_table.py gpurun_out/r06_pmc_$cfg.json gpurun_out/r06_rocprofv3_kernel_stats_$cfg.csv > gpurun_out/r06_pmc_$cfg.txt
  rm -rf gpurun_out/pmc_${cfg}_FETCH_SIZE gpurun_out/pmc_${cfg}_WRITE_SIZE
fi
head -c 600 gpurun_out/r06_bench_$cfg.json; echo
