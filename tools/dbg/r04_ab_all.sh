# A/B of the working library against paropt_amd/libbase.so (a committed state built from a worktree), all four
# configurations alternating in ONE call, then the GPU suite.   usage: bash tools/dbg/r04_ab_all.sh <tag>
set -u
tag=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
c4="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin --steps 20 --repeats 3"
c3="--steps 20 --warmup 5 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 3"
c2="--nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 --boundary builtin --no-cpu-baseline --skip-extension-variant --repeats 3"
for round in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export PAROPT_AMD_LIB=$PWD/paropt_amd/libbase.so; else unset PAROPT_AMD_LIB; fi
    python3 tools/bench_tr.py --no-cpu-baseline > gpurun_out/r04_ab_${tag}_c5_${lib}_$round.json 2>>gpurun_out/r04_ab_$tag.err
    python3 bench.py $c4 > gpurun_out/r04_ab_${tag}_c4_${lib}_$round.json 2>>gpurun_out/r04_ab_$tag.err
    python3 bench.py $c2 > gpurun_out/r04_ab_${tag}_c2_${lib}_$round.json 2>>gpurun_out/r04_ab_$tag.err
    python3 bench.py $c3 > gpurun_out/r04_ab_${tag}_c3_${lib}_$round.json 2>>gpurun_out/r04_ab_$tag.err
  done
done
unset PAROPT_AMD_LIB
python3 - <<EOF
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_ab_${tag}_c*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "failed", e); continue
    if "inner_ip_iterations_per_s" in d:
        print(f.split("/")[-1], "%.2f TR it/s" % d["value"], "%.0f inner it/s" % d["inner_ip_iterations_per_s"], d["inner_ip_iterations"])
    else:
        print(f.split("/")[-1], "%.2f it/s" % d["value"], "%.3f ms" % d["ms_per_step"])
EOF
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04_gputests_$tag.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r04_gputests_$tag.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r04_gputests_$tag.log | head
