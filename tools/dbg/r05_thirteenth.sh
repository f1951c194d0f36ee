#!/bin/bash
# round 5, thirteenth GPU call: the first solve pass two tiles per step (solve2_dots2_kernel): bit-identity test, the
# interior-point and trust-region suites, then A/B inside one process each (debug switch 13) at configs 5, 4, 2
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_ip.py -m gpu -q --no-header -x -k "two_tiles" 2>&1 | tail -8
python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -15 > gpurun_out/r05_tests13.log
tail -3 gpurun_out/r05_tests13.log
Q="--no-cpu-baseline --repeats 3 --skip-extension-variant --boundary builtin"
run() {  # tag, two, args...
  tag=$1; two=$2; shift 2
  PAROPT_AMD_S2D_TWO=$two python3 "$@" 2> gpurun_out/r05_ab13_$tag.err | grep '"metric"' > gpurun_out/r05_ab13_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_ab13_%s.json" % tag).read())
    ph = d.get("phase_ms_per_iter") or {}
    print(tag, "value %.3f" % d["value"], "ms %.4f" % (d.get("ms_per_step") or d.get("ms_per_inner_iteration") or 0.0),
          "inner %s" % d.get("inner_ip_iterations_per_s"), "inner its %s" % d.get("inner_ip_iterations"), "kkt_step %.3f" % ph.get("kkt_step", 0.0))
except Exception as e:
    print(tag, "FAILED", e)
PY
}
for rep in 1 2; do
  run c5_one$rep 0 tools/bench_tr.py --no-cpu-baseline --repeats 3
  run c5_two$rep 1 tools/bench_tr.py --no-cpu-baseline --repeats 3
  run c4_one$rep 0 bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
  run c4_two$rep 1 bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
done
for cfg in c5 c4; do
  if [ $cfg = c5 ]; then prog=tools/bench_tr.py; args="--no-cpu-baseline --repeats 1"; else prog=bench.py; args="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin"; fi
  rm -rf gpurun_out/prof13_$cfg
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof13_$cfg -o $cfg --output-format csv -- python3 $prog $args > /dev/null 2> gpurun_out/prof13_$cfg.err
  cp gpurun_out/prof13_$cfg/*kernel_stats.csv gpurun_out/r05_two_tiles_kernel_stats_$cfg.csv
  rm -rf gpurun_out/prof13_$cfg
  grep "solve2_dots" gpurun_out/r05_two_tiles_kernel_stats_$cfg.csv | cut -c1-60,200-400 | head -6
done
