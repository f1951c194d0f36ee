#!/bin/bash
# round 5, eighth GPU call: (1) whole GPU suite on the build with branch-free producers + the 64-row wide-panel Gram +
# ParOptInfeasSubproblem; (2) A/B in ONE call against the library of the commit before (paropt_amd/libparopt_amd_prev.so,
# same sources except wgram.hip): configs 3, 4, 2, config 3 with L-BFGS(20) (73-column Gram), config 5
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -15 > gpurun_out/r05_tests8.log
tail -5 gpurun_out/r05_tests8.log
Q="--no-cpu-baseline --repeats 3 --skip-extension-variant --boundary builtin"
run() {  # tag, lib, args...
  tag=$1; libf=$2; shift 2
  PAROPT_AMD_LIB=$libf python3 "$@" 2> gpurun_out/r05_ab8_$tag.err | grep '"metric"' > gpurun_out/r05_ab8_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
try:
    d = json.loads(open("gpurun_out/r05_ab8_%s.json" % tag).read())
    ph = d.get("phase_ms_per_iter") or {}
    print(tag, "value %.3f" % d["value"], "ms %.4f" % (d.get("ms_per_step") or d.get("ms_per_inner_iteration") or 0.0),
          "inner %s" % d.get("inner_ip_iterations_per_s"), "setup_kkt %.3f" % ph.get("setup_kkt", 0.0))
except Exception as e:
    print(tag, "FAILED", e)
PY
}
NEW=$PWD/paropt_amd/libparopt_amd.so
OLD=$PWD/paropt_amd/libparopt_amd_prev.so
for rep in 1 2; do
  run c3_old$rep $OLD bench.py --steps 20 --warmup 5 $Q
  run c3_new$rep $NEW bench.py --steps 20 --warmup 5 $Q
done
run c4_old $OLD bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
run c4_new $NEW bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
run c4_old2 $OLD bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
run c4_new2 $NEW bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
run c2_old $OLD bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 $Q
run c2_new $NEW bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 $Q
run c3l_old $OLD bench.py --qn bfgs --qn-size 20 --steps 20 --warmup 22 $Q
run c3l_new $NEW bench.py --qn bfgs --qn-size 20 --steps 20 --warmup 22 $Q
run c5_old $OLD tools/bench_tr.py --no-cpu-baseline --repeats 3
run c5_new $NEW tools/bench_tr.py --no-cpu-baseline --repeats 3
