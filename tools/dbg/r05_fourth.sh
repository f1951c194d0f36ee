#!/bin/bash
# round 5, fourth GPU call: full GPU suite, A/B of the producer-side group sums at config 4, config 4 line of record
# with kernel stats + PMC, config 3's L-BFGS(20) run to its tolerance
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests -m gpu -q --no-header 2>&1 | tail -40 > gpurun_out/r05_gputests_full2.log
tail -6 gpurun_out/r05_gputests_full2.log
python3 tools/ab_switch.py --variants "12=0;12=1" --rounds 4 --what iter --n 20000000 --c 4 --k 10 --qn bfgs --nwcon 1000000 --nw 20 > gpurun_out/r05_ab_gs_producer.jsonl 2> gpurun_out/r05_ab_gs_producer.err
grep -h "ms_per_iter\|wgram_launch" gpurun_out/r05_ab_gs_producer.jsonl | cut -c1-220
bash tools/collect_r05.sh c4 > gpurun_out/r05_collect_c4.log 2>&1
tail -2 gpurun_out/r05_collect_c4.log | cut -c1-400
python3 tools/bench_convergent.py --n 50000000 --qn-size 20 --tol 1e-6 --max-iters 8000 --repeats 1 --no-cpu > gpurun_out/r05_convergent_c3_lbfgs20_gpu_full.json 2> gpurun_out/r05_convergent_full.err
head -c 900 gpurun_out/r05_convergent_c3_lbfgs20_gpu_full.json
