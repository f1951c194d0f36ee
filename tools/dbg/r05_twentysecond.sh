#!/bin/bash
# round 5: the trust-region sweep's suite cases, then a campaign of 120 (bounded: the oracle runs on the box's CPU)
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_tr_sweep.py -m gpu -q --no-header 2>&1 | tail -6 | cut -c1-400
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
( echo "# trust-region driver against oracle/tr_oracle.py: PAROPT_TR_SWEEP_CASES=120 PAROPT_TR_SWEEP_SEED=6262"
  PAROPT_TR_SWEEP_CASES=120 PAROPT_TR_SWEEP_SEED=6262 timeout 900 python3 tests/test_gpu_tr_sweep.py 2>&1 | grep -v "$F" | cut -c1-1500
) > gpurun_out/r05_tr_sweep_campaign.txt
grep "differ\|ERROR\|^TR CASE" gpurun_out/r05_tr_sweep_campaign.txt | cut -c1-300
