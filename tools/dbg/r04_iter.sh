# One development round on the GPU box: GPU test suite, config 4 at full and per-rank-of-4 size, launch sequence.
#   usage: bash tools/dbg/r04_iter.sh <tag> [pytest args]
set -u
tag=${1:-x}
shift
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -m gpu -q "$@" > gpurun_out/r04_gputests_$tag.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r04_gputests_$tag.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r04_gputests_$tag.log | head -20
c4="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin"
python3 bench.py $c4 --steps 20 --repeats 3 > gpurun_out/r04_c4_$tag.json 2>gpurun_out/r04_c4_$tag.err
python3 bench.py --nglobal 5000000 --ncon 4 --nwcon 250000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin --steps 20 --repeats 3 > gpurun_out/r04_c4q_$tag.json 2>/dev/null
rm -rf gpurun_out/trace_c4
rocprofv3 --kernel-trace -d gpurun_out/trace_c4 -o t --output-format csv -- python3 bench.py $c4 --steps 4 --repeats 1 > /dev/null 2> gpurun_out/trace_c4.err
python3 tools/dbg/launch_seq.py gpurun_out/trace_c4 420 > gpurun_out/r04_launch_seq_c4_$tag.txt
rm -rf gpurun_out/trace_c4
python3 - <<EOF
import json
for f in ("r04_c4_$tag.json", "r04_c4q_$tag.json"):
    try:
        d = json.load(open("gpurun_out/" + f))
        print(f, "%.1f it/s" % d["value"], "%.3f ms" % d["ms_per_step"], "frac %.3f" % d["iteration_frac"],
              "launches %.0f" % d["config"]["launches_per_iter"], "syncs %.0f" % d["config"]["reductions_per_iter"])
        print("  ", {k: round(v, 3) for k, v in d["phase_ms_per_iter"].items()})
    except Exception as e:
        print(f, "failed", e)
EOF
