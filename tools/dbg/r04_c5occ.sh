set -u
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
for round in 1 2; do
  python3 tools/bench_tr.py --no-cpu-baseline > gpurun_out/r04_c5occ_def_$round.json 2>>gpurun_out/r04_c5occ.err
  PAROPT_AMD_S2D_OCC=2 python3 tools/bench_tr.py --no-cpu-baseline > gpurun_out/r04_c5occ_occ2_$round.json 2>>gpurun_out/r04_c5occ.err
  PAROPT_AMD_LIB=$PWD/paropt_amd/libbase.so python3 tools/bench_tr.py --no-cpu-baseline > gpurun_out/r04_c5occ_base_$round.json 2>>gpurun_out/r04_c5occ.err
done
python3 - <<EOF
import json, glob
for f in sorted(glob.glob("gpurun_out/r04_c5occ_*.json")):
    try:
        d = json.load(open(f))
        print(f.split("/")[-1], "%.2f TR it/s" % d["value"], "%.0f inner it/s" % d["inner_ip_iterations_per_s"], "inner", d["inner_ip_iterations"],
              "launches %.1f" % d["launches_per_inner_iteration"], "syncs %.2f" % d["host_syncs_per_inner_iteration"])
    except Exception as e:
        print(f, "failed", e)
EOF
