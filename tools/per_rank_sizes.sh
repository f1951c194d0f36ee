#!/bin/bash
# The per-rank workload of the strong-scaling metric at 1 / 2 / 4 / 8 ranks on ONE GPU, without any collective
# (config 3: n = 50 M / N; config 4: n = 20 M / N, w = 1 M / N): what a rank of an N-GPU run has to do per iteration.
#   usage (repo root on the GPU box):  bash tools/per_rank_sizes.sh > gpurun_out/r04_per_rank_sizes.txt
cd /tmp && export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "# bash tools/per_rank_sizes.sh (one gpurun call, one GPU): bench.py --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 3"
for N in 1 2 4 8; do
  python3 bench.py --nglobal $((50000000 / N)) --steps 20 --warmup 5 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])  # (RCCL prints its banner behind the line)
print('config 3  n/$N  %8.3f ms per iteration  %7.1f it/s  %.0f host syncs, %.0f launches per iteration, iteration_frac %.3f' % (d['ms_per_step'], d['value'], d['config']['reductions_per_iter'], d['config']['launches_per_iter'], d['iteration_frac']))"
done
for N in 1 2 4; do
  python3 bench.py --nglobal $((20000000 / N)) --ncon 4 --nwcon $((1000000 / N)) --nw 20 --qn bfgs --steps 20 --warmup 12 --no-cpu-baseline --skip-extension-variant --boundary builtin --repeats 3 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin.read().strip().splitlines() if l.startswith('{')][-1])  # (RCCL prints its banner behind the line)
print('config 4  n/$N  %8.3f ms per iteration  %7.1f it/s  %.0f host syncs, %.0f launches per iteration, iteration_frac %.3f' % (d['ms_per_step'], d['value'], d['config']['reductions_per_iter'], d['config']['launches_per_iter'], d['iteration_frac']))"
done
