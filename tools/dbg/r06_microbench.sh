# every hot kernel in isolation on the final build: the metric's shape, config 2's, config 5's steering / subproblem shapes
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
rm -f gpurun_out/r06_microbench.jsonl
python tools/microbench.py --tag "config 3 shape" >> gpurun_out/r06_microbench.jsonl
python tools/microbench.py --n 10000000 --c 8 --k 12 --tag "config 2 shape (k capped at 12)" >> gpurun_out/r06_microbench.jsonl
python tools/microbench.py --n 5000000 --c 4 --k 0 --tag "config 5 steering shape" >> gpurun_out/r06_microbench.jsonl
python tools/microbench.py --n 5000000 --c 4 --k 10 --tag "config 5 subproblem shape" >> gpurun_out/r06_microbench.jsonl
grep -c kernel gpurun_out/r06_microbench.jsonl
grep "corr\|solve2c\|kkt_res_update\|comp_merit\|solve2(first)\|\"d1\"\|mdot(c)" gpurun_out/r06_microbench.jsonl | cut -c1-260
