set -u
mkdir -p gpurun_out/r03g
(timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -8) > gpurun_out/r03g/gputests.log
python bench.py --steps 20 --warmup 12 --repeats 3 --no-cpu-baseline --boundary builtin > gpurun_out/r03g/bench.json 2>/dev/null
PAROPT_AMD_NO_FUSED_MERIT=1 python bench.py --steps 20 --warmup 12 --repeats 3 --no-cpu-baseline --boundary builtin --skip-extension-variant > gpurun_out/r03g/bench_nofuse.json 2>/dev/null
cat gpurun_out/r03g/gputests.log
python - <<'PY'
import json
for f in ("bench","bench_nofuse"):
    r=json.load(open("gpurun_out/r03g/%s.json"%f))
    print(f, {k:(round(v["value"],2), round(v["ms_per_step"],3), v["host_syncs_per_iter"], v["launches_per_iter"], round(v["iteration_frac"],3)) for k,v in r["variants"].items()})
    print(r["phase_ms_per_iter"])
PY
