#!/bin/bash
# round 5: the tiled group kernels with requests two tiles ahead: sparse-constraint tests, then A/B at config 4 against
# the library with the previous wcon.hip (paropt_amd/libparopt_amd_prev.so), kernel statistics
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_vec.py tests/test_gpu_user_problem.py tests/test_gpu_ip.py tests/test_gpu_random_sweep.py -m gpu -q --no-header -x 2>&1 | tail -4
Q="--no-cpu-baseline --repeats 3 --skip-extension-variant --boundary builtin"
run() {
  tag=$1; libf=$2; shift 2
  PAROPT_AMD_LIB=$libf python3 "$@" 2> gpurun_out/r05_abg_$tag.err | grep '"metric"' > gpurun_out/r05_abg_$tag.json
  python3 - "$tag" <<'PY'
import json, sys
tag = sys.argv[1]
d = json.loads(open("gpurun_out/r05_abg_%s.json" % tag).read())
print(tag, "value %.3f" % d["value"], "ms %.4f" % d["ms_per_step"])
PY
}
NEW=$PWD/paropt_amd/libparopt_amd.so
OLD=$PWD/paropt_amd/libparopt_amd_prev.so
for rep in 1 2 3; do
  run c4_old$rep $OLD bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
  run c4_new$rep $NEW bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 $Q
done
rm -rf gpurun_out/profg
rocprofv3 --kernel-trace --stats -d gpurun_out/profg -o c4 --output-format csv -- python3 bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 --no-cpu-baseline --repeats 1 --skip-extension-variant --boundary builtin > /dev/null 2> gpurun_out/profg.err
cp gpurun_out/profg/*kernel_stats.csv gpurun_out/r05_group_prefetch_kernel_stats_c4.csv
rm -rf gpurun_out/profg
grep "group_" gpurun_out/r05_group_prefetch_kernel_stats_c4.csv | cut -c1-50,100-260 | head
