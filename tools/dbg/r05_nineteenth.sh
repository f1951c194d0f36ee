#!/bin/bash
# round 5, nineteenth GPU call: the large-n draws (several tiles per workgroup in every persistent kernel): the suite's
# 12 and a campaign of 150
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_random_sweep.py -m gpu -q --no-header -k "large" 2>&1 | tail -5
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
( echo "# large-n draws (LARGE_NS = 32 769 ... 393 217): PAROPT_SWEEP_CASES=1 PAROPT_SWEEP_LARGE_CASES=150 PAROPT_SWEEP_SEED=4444 PAROPT_SWEEP_QN_CASES=1"
  PAROPT_SWEEP_CASES=1 PAROPT_SWEEP_LARGE_CASES=150 PAROPT_SWEEP_SEED=4444 PAROPT_SWEEP_QN_CASES=1 timeout 3000 python3 tests/test_gpu_random_sweep.py 2>&1 | grep -v "$F" | cut -c1-1200
) > gpurun_out/r05_sweep_campaign_large.txt
grep "differ\|ERROR\|^LARGE" gpurun_out/r05_sweep_campaign_large.txt | cut -c1-300
