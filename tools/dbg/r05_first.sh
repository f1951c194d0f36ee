#!/bin/bash
# round 5, first GPU call: full GPU suite, per-iteration trajectory errors, vector-API roofline rows
mkdir -p gpurun_out
python -m pytest tests -m gpu -q --no-header -x 2>&1 | tail -40 > gpurun_out/r05_gputests.log
tail -5 gpurun_out/r05_gputests.log
python tools/trajectory_errors.py > gpurun_out/r05_trajectory_errors.log 2>&1
tail -3 gpurun_out/r05_trajectory_errors.log
python tools/microbench.py --vec-api --n 50000000 --reps 10 --tag r05 > gpurun_out/r05_microbench_vec_50M.jsonl 2> gpurun_out/r05_microbench_vec_50M.err
python tools/microbench.py --vec-api --n 10000000 --reps 20 --tag r05 > gpurun_out/r05_microbench_vec_10M.jsonl 2> gpurun_out/r05_microbench_vec_10M.err
tail -3 gpurun_out/r05_microbench_vec_50M.jsonl gpurun_out/r05_microbench_vec_50M.err
