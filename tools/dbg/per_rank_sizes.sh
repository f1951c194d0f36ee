set -u
mkdir -p gpurun_out/r03p
for d in 1 2 4 8; do
  n=$((50000000 / d))
  python bench.py --nglobal $n --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 3 > gpurun_out/r03p/n_over_$d.json 2>/dev/null
done
PAROPT_BENCH_SHARE_GPU=1 python bench.py --gpus 2 --nglobal 8000000 --steps 10 --warmup 12 --repeats 2 --no-cpu-baseline --boundary builtin --skip-extension-variant > gpurun_out/r03p/two_ranks_shared.json 2>/dev/null
PAROPT_BENCH_SHARE_GPU=1 python bench.py --gpus 8 --nglobal 8000003 --steps 10 --warmup 12 --repeats 2 --no-cpu-baseline --boundary builtin --skip-extension-variant > gpurun_out/r03p/eight_ranks_shared.json 2>/dev/null
python - <<'PY'
import json
print("# python bench.py --nglobal N --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 3 on ONE GPU, one gpurun call:")
print("# the per-rank workload of the strong-scaling metric (reference problem contract) at 1 / 2 / 4 / 8 ranks without any collective")
for d in (1,2,4,8):
    r=json.load(open("gpurun_out/r03p/n_over_%d.json"%d))
    print("n/%d  %8.3f ms per iteration  %7.1f it/s  %.1f host syncs, %.1f launches per iteration, iteration_frac %.3f" % (d, r["ms_per_step"], r["value"], r["config"]["reductions_per_iter"], r["config"]["launches_per_iter"], r["iteration_frac"]))
for f in ("two_ranks_shared","eight_ranks_shared"):
    r=json.load(open("gpurun_out/r03p/%s.json"%f))
    print(f, "n_gpus", r["n_gpus"], "n_local", r["config"]["n_local"], "ms/iter %.3f"%r["ms_per_step"], "collective", r["config"]["collective"], "collective_us", r["collective_us"])
PY
