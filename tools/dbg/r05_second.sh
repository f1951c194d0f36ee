#!/bin/bash
# round 5, second GPU call: the exported config-4 route, the tolerance schedule, vector-API rows with kernel-only times
mkdir -p gpurun_out
python -m pytest tests/test_gpu_user_problem.py tests/test_gpu_csr.py tests/test_gpu_compat.py tests/test_gpu_ip.py tests/test_gpu_kat.py -q --no-header -x 2>&1 | tail -40 > gpurun_out/r05_tests2.log
tail -8 gpurun_out/r05_tests2.log
python tools/microbench.py --vec-api --n 50000000 --reps 10 --tag r05 > gpurun_out/r05_microbench_vec_50M.jsonl 2> gpurun_out/r05_microbench_vec_50M.err
python tools/microbench.py --vec-api --n 10000000 --reps 20 --tag r05 > gpurun_out/r05_microbench_vec_10M.jsonl 2> gpurun_out/r05_microbench_vec_10M.err
tail -2 gpurun_out/r05_microbench_vec_50M.err
python bench.py --nglobal 20000000 --ncon 4 --nwcon 1000000 --nw 20 --qn bfgs --steps 20 --warmup 12 --boundary both --skip-extension-variant --no-cpu-baseline > gpurun_out/r05_bench_c4_boundary.json 2> gpurun_out/r05_bench_c4_boundary.err
tail -3 gpurun_out/r05_bench_c4_boundary.err
head -c 600 gpurun_out/r05_bench_c4_boundary.json
