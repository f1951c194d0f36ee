set -u
mkdir -p gpurun_out
python bench.py --steps 20 --warmup 5 > gpurun_out/bench_r03_final.json 2> gpurun_out/bench_r03_final.err
python tools/microbench.py --tag r03 --reps 5 > gpurun_out/r03_microbench.jsonl 2>/dev/null
bash tools/collect_profiles.sh > gpurun_out/collect_profiles.log 2>&1
bash tools/collect_pmc.sh r03 > gpurun_out/collect_pmc.log 2>&1
tail -3 gpurun_out/collect_profiles.log; tail -3 gpurun_out/collect_pmc.log
head -c 300 gpurun_out/bench_r03_final.json
