#!/bin/bash
# round 5, ninth GPU call: the first solve pass with its column and element-operand requests issued one tile ahead
# (solve2_dots_kernel, EARLY): whole GPU suite (bits must not move), then A/B in one call against the same sources
# built with -DPO_S2D_LATE_PREFETCH (paropt_amd/libparopt_amd_late.so), then kernel statistics of configs 5 and 4
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -15 > gpurun_out/r05_tests9.log
tail -3 gpurun_out/r05_tests9.log
Q="--no-cpu-baseline --repeats 3 --skip-extension-variant --boundary builtin"
run() {  # tag, lib, args...
  tag=$1; libf=$2; shift 2
  PAROPT_AMD_LIB=$libf python3 "$@" 2> gpurun_out/r05_ab9_$tag.err | grep '"metric"' > gpurun_out/r05_ab9_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_ab9_%s.json" % tag).read())
    ph = d.get("phase_ms_per_iter") or {}
    print(tag, "value %.3f" % d["value"], "ms %.4f" % (d.get("ms_per_step") or d.get("ms_per_inner_iteration") or 0.0),
          "inner %s" % d.get("inner_ip_iterations_per_s"), "kkt_step %.3f" % ph.get("kkt_step", 0.0))
except Exception as e:
    print(tag, "FAILED", e)
PY
}
NEW=$PWD/paropt_amd/libparopt_amd.so
OLD=$PWD/paropt_amd/libparopt_amd_late.so
for rep in 1 2; do
  run c5_late$rep $OLD tools/bench_tr.py --no-cpu-baseline --repeats 3
  run c5_early$rep $NEW tools/bench_tr.py --no-cpu-baseline --repeats 3
  run c4_late$rep $OLD bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
  run c4_early$rep $NEW bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
  run c3_late$rep $OLD bench.py --steps 20 --warmup 5 $Q
  run c3_early$rep $NEW bench.py --steps 20 --warmup 5 $Q
  run c2_late$rep $OLD bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 $Q
  run c2_early$rep $NEW bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 $Q
done
run c3l_late $OLD bench.py --qn bfgs --qn-size 20 --steps 20 --warmup 22 $Q
run c3l_early $NEW bench.py --qn bfgs --qn-size 20 --steps 20 --warmup 22 $Q
for cfg in c5 c4; do
  if [ $cfg = c5 ]; then prog=tools/bench_tr.py; args="--no-cpu-baseline --repeats 1"; else prog=bench.py; args="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin"; fi
  rm -rf gpurun_out/prof9_$cfg
  rocprofv3 --kernel-trace --stats -d gpurun_out/prof9_$cfg -o $cfg --output-format csv -- python3 $prog $args > /dev/null 2> gpurun_out/prof9_$cfg.err
  cp gpurun_out/prof9_$cfg/*kernel_stats.csv gpurun_out/r05_early_kernel_stats_$cfg.csv
  rm -rf gpurun_out/prof9_$cfg
  grep "solve2_dots" gpurun_out/r05_early_kernel_stats_$cfg.csv | cut -c1-60,200-400 | head -6
done
