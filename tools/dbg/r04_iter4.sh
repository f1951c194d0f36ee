# Development round: A/B of the working library against paropt_amd/libbase.so (built from a committed state) in ONE call:
# config 4 and config 3 alternately, kernel stats of config 4, then the GPU suite.
set -u
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
c4="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin --steps 20 --repeats 3"
c3="--steps 20 --warmup 5 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 3"
for round in 1 2; do
  python3 bench.py $c4 > gpurun_out/r04_ab_${tag}_c4_new_$round.json 2>>gpurun_out/r04_ab_$tag.err
  PAROPT_AMD_LIB=$PWD/paropt_amd/libbase.so python3 bench.py $c4 > gpurun_out/r04_ab_${tag}_c4_base_$round.json 2>>gpurun_out/r04_ab_$tag.err
  python3 bench.py $c3 > gpurun_out/r04_ab_${tag}_c3_new_$round.json 2>>gpurun_out/r04_ab_$tag.err
  PAROPT_AMD_LIB=$PWD/paropt_amd/libbase.so python3 bench.py $c3 > gpurun_out/r04_ab_${tag}_c3_base_$round.json 2>>gpurun_out/r04_ab_$tag.err
done
rm -rf gpurun_out/trace_c4
rocprofv3 --kernel-trace --stats -d gpurun_out/trace_c4 -o t --output-format csv -- python3 bench.py $c4 --repeats 1 > /dev/null 2> gpurun_out/trace_c4.err
python3 tools/dbg/launch_seq.py gpurun_out/trace_c4 400 > gpurun_out/r04_launch_seq_c4_$tag.txt
cp gpurun_out/trace_c4/*kernel_stats.csv gpurun_out/r04_kernel_stats_c4_$tag.csv 2>/dev/null
rm -rf gpurun_out/trace_c4
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04_gputests_$tag.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r04_gputests_$tag.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r04_gputests_$tag.log | head -20
python3 - <<EOF
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab_${tag}_c*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "%.2f it/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["iteration_frac"],
              "launches %.0f" % d["config"]["launches_per_iter"], "GB %.2f" % (d["iteration_bytes"]/1e9))
    except Exception as e:
        print(f, "failed", e)
EOF
