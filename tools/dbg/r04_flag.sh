set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
c4="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin --steps 20 --repeats 3"
c2="--nglobal 10000000 --ncon 8 --qn bfgs --qn-size 20 --problem quadratic --steps 20 --warmup 22 --boundary builtin --no-cpu-baseline --skip-extension-variant --repeats 3"
for round in 1 2; do
  for nf in 1 0; do
    if [ $nf = 1 ]; then export PAROPT_AMD_NO_FLAG_POLL=1; else unset PAROPT_AMD_NO_FLAG_POLL; fi
    python3 tools/bench_tr.py --no-cpu-baseline > gpurun_out/r04_flag_c5_nf${nf}_$round.json 2>>gpurun_out/r04_flag.err
    python3 bench.py $c4 > gpurun_out/r04_flag_c4_nf${nf}_$round.json 2>>gpurun_out/r04_flag.err
    python3 bench.py $c2 > gpurun_out/r04_flag_c2_nf${nf}_$round.json 2>>gpurun_out/r04_flag.err
  done
done
unset PAROPT_AMD_NO_FLAG_POLL
python3 - <<EOF
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_flag_c*.json")):
    try:
        d = json.load(open(f))
    except Exception as e:
        print(f, "failed", e); continue
    if "inner_ip_iterations_per_s" in d:
        print(f.split("/")[-1], "%.2f TR it/s" % d["value"], "%.0f inner it/s" % d["inner_ip_iterations_per_s"], d["inner_ip_iterations"])
    else:
        print(f.split("/")[-1], "%.2f it/s" % d["value"], "%.3f ms" % d["ms_per_step"])
EOF
timeout 1500 python -m pytest tests -m gpu -q -x > gpurun_out/r04_gputests_flag.log 2>&1; echo "pytest rc $?"
grep -E "passed|failed|error" gpurun_out/r04_gputests_flag.log | tail -3
grep -E "^FAILED|^ERROR" gpurun_out/r04_gputests_flag.log | head
