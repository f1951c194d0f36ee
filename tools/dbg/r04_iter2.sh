# Development round on the GPU box: GPU test suite, configs 4 and 5 (bench lines + launch sequences).
#   usage: bash tools/dbg/r04_iter2.sh <tag> [pytest args]
set -u
tag=${1:-x}
shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q "$@" > gpurun_out/r04_gputests_$tag.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r04_gputests_$tag.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r04_gputests_$tag.log | head -20
c4="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin"
python3 bench.py $c4 --steps 20 --repeats 3 > gpurun_out/r04_c4_$tag.json 2>gpurun_out/r04_c4_$tag.err
python3 tools/bench_tr.py --no-cpu-baseline > gpurun_out/r04_c5_$tag.json 2>gpurun_out/r04_c5_$tag.err
python3 tools/bench_tr.py --no-cpu-baseline --assembly objects > gpurun_out/r04_c5obj_$tag.json 2>>gpurun_out/r04_c5_$tag.err
rm -rf gpurun_out/trace_c5
rocprofv3 --kernel-trace --stats -d gpurun_out/trace_c5 -o t --output-format csv -- python3 tools/bench_tr.py --no-cpu-baseline --repeats 1 > /dev/null 2> gpurun_out/trace_c5.err
python3 tools/dbg/launch_seq.py gpurun_out/trace_c5 3000 > gpurun_out/r04_launch_seq_c5_$tag.txt
cp gpurun_out/trace_c5/*kernel_stats.csv gpurun_out/r04_kernel_stats_c5_$tag.csv 2>/dev/null
rm -rf gpurun_out/trace_c5
python3 - <<EOF
import json
for f in ("r04_c4_$tag.json",):
    try:
        d = json.load(open("gpurun_out/" + f))
        print(f, "%.1f it/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["iteration_frac"],
              "launches %.0f" % d["config"]["launches_per_iter"], "syncs %.0f" % d["config"]["reductions_per_iter"])
    except Exception as e:
        print(f, "failed", e)
for f in ("r04_c5_$tag.json", "r04_c5obj_$tag.json"):
    try:
        d = json.load(open("gpurun_out/" + f))
        print(f, "%.2f TR it/s" % d["value"], "%.0f inner it/s" % d["inner_ip_iterations_per_s"], "inner", d["inner_ip_iterations"],
              "launches %.1f" % d["launches_per_inner_iteration"], "syncs %.2f" % d["host_syncs_per_inner_iteration"],
              "frac %.3f" % d["iteration_frac"], "mdot", d["roofline"] and round(d["roofline"]["frac"], 3))
    except Exception as e:
        print(f, "failed", e)
EOF
