#!/bin/bash
# round 5, twentieth GPU call: the trust-region random sweep (suite cases, then a campaign of 300)
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
python -m pytest tests/test_gpu_tr_sweep.py -m gpu -q --no-header 2>&1 | tail -12 | cut -c1-600
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
( echo "# trust-region driver against oracle/tr_oracle.py: PAROPT_TR_SWEEP_CASES=300 PAROPT_TR_SWEEP_SEED=5151"
  PAROPT_TR_SWEEP_CASES=300 PAROPT_TR_SWEEP_SEED=5151 timeout 3000 python3 tests/test_gpu_tr_sweep.py 2>&1 | grep -v "$F" | cut -c1-1500
) > gpurun_out/r05_tr_sweep_campaign.txt
grep "differ\|ERROR\|^TR CASE" gpurun_out/r05_tr_sweep_campaign.txt | cut -c1-400
