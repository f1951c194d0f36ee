set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
c4="--nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin --steps 20 --repeats 3"
for round in 1 2 3; do
  python3 bench.py $c4 > gpurun_out/r04_s2dg_on_$round.json 2>>gpurun_out/r04_s2dg.err
  PAROPT_AMD_GROUP_COLS_S2D=0 python3 bench.py $c4 > gpurun_out/r04_s2dg_off_$round.json 2>>gpurun_out/r04_s2dg.err
done
python3 - <<EOF
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_s2dg_*.json")):
    d = json.load(open(f))
    print(f.split("/")[-1], "%.2f it/s" % d["value"], "%.3f ms" % d["ms_per_step"], "launches %.0f" % d["config"]["launches_per_iter"], "GB %.2f" % (d["iteration_bytes"]/1e9), d["phase_ms_per_iter"]["kkt_step"])
EOF
