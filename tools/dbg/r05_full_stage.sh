#!/bin/bash
# round 5: Gram producers without the per-lane zero selects on full tiles: wgram tests, then A/B in one call against the
# same sources built with -DPO_WGRAM_NO_FULL_STAGE
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_vec.py tests/test_gpu_kat.py -m gpu -q --no-header -x 2>&1 | tail -2
Q="--no-cpu-baseline --repeats 3 --skip-extension-variant --boundary builtin"
run() {
  tag=$1; libf=$2; shift 2
  PAROPT_AMD_LIB=$libf python3 "$@" 2> gpurun_out/r05_abf_$tag.err | grep '"metric"' > gpurun_out/r05_abf_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads(open("gpurun_out/r05_abf_%s.json" % tag).read())
ph = d.get("phase_ms_per_iter") or {}
print(tag, "value %.3f" % d["value"], "ms %.4f" % d["ms_per_step"], "setup_kkt %.3f" % ph.get("setup_kkt", 0.0))
PY
}
NEW=$PWD/paropt_amd/libparopt_amd.so
OLD=$PWD/paropt_amd/libparopt_amd_prev.so
for rep in 1 2 3; do
  run c3_old$rep $OLD bench.py --steps 20 --warmup 5 $Q
  run c3_new$rep $NEW bench.py --steps 20 --warmup 5 $Q
done
for rep in 1 2; do
  run c2_old$rep $OLD bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 $Q
  run c2_new$rep $NEW bench.py --nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 $Q
  run c4_old$rep $OLD bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
  run c4_new$rep $NEW bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
done
run c3l_old $OLD bench.py --qn bfgs --qn-size 20 --steps 20 --warmup 22 $Q
run c3l_new $NEW bench.py --qn bfgs --qn-size 20 --steps 20 --warmup 22 $Q
