# Development round: focused tests first, then the GPU suite, config 4 bench (+ optional A/B of the fused panel image)
set -u
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_vec.py -m gpu -q -k "structured_panel or declines" > gpurun_out/r04_focus_$tag.log 2>&1; echo "focus rc $?"
tail -5 gpurun_out/r04_focus_$tag.log
c4="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin"
python3 bench.py $c4 --steps 20 --repeats 3 > gpurun_out/r04_c4_$tag.json 2>gpurun_out/r04_c4_$tag.err
PAROPT_AMD_NO_GRAM_GROUPS=1 python3 bench.py $c4 --steps 20 --repeats 3 > gpurun_out/r04_c4_${tag}_nogroups.json 2>>gpurun_out/r04_c4_$tag.err
rm -rf gpurun_out/trace_c4
rocprofv3 --kernel-trace --stats -d gpurun_out/trace_c4 -o t --output-format csv -- python3 bench.py $c4 --steps 20 --repeats 1 > /dev/null 2> gpurun_out/trace_c4.err
python3 tools/dbg/launch_seq.py gpurun_out/trace_c4 400 > gpurun_out/r04_launch_seq_c4_$tag.txt
cp gpurun_out/trace_c4/*kernel_stats.csv gpurun_out/r04_kernel_stats_c4_$tag.csv 2>/dev/null
rm -rf gpurun_out/trace_c4
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04_gputests_$tag.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r04_gputests_$tag.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r04_gputests_$tag.log | head -20
python3 - <<EOF
import json
for f in ("r04_c4_$tag.json", "r04_c4_${tag}_nogroups.json"):
    try:
        d = json.load(open("gpurun_out/" + f))
        print(f, "%.1f it/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["iteration_frac"],
              "launches %.0f" % d["config"]["launches_per_iter"], "syncs %.0f" % d["config"]["reductions_per_iter"], "GB %.2f" % (d["iteration_bytes"]/1e9))
    except Exception as e:
        print(f, "failed", e)
EOF
