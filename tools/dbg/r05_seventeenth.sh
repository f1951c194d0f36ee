#!/bin/bash
# round 5, seventeenth GPU call: three more campaigns (two fresh seeds of 1500 draws, one of 600 through the RCCL
# communicator) with 200 quasi-Newton sequences each
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-/root/repo}"
F="^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|general CSR path"
( echo "# PAROPT_SWEEP_CASES=1500 PAROPT_SWEEP_SEED=1111 PAROPT_SWEEP_QN_CASES=200"
  PAROPT_SWEEP_CASES=1500 PAROPT_SWEEP_SEED=1111 PAROPT_SWEEP_QN_CASES=200 timeout 1500 python3 tests/test_gpu_random_sweep.py 2>&1 | grep -v "$F" | cut -c1-1200
  echo
  echo "# PAROPT_SWEEP_CASES=1500 PAROPT_SWEEP_SEED=2222 PAROPT_SWEEP_QN_CASES=200"
  PAROPT_SWEEP_CASES=1500 PAROPT_SWEEP_SEED=2222 PAROPT_SWEEP_QN_CASES=200 timeout 1500 python3 tests/test_gpu_random_sweep.py 2>&1 | grep -v "$F" | cut -c1-1200
  echo
  echo "# every reduction through ncclAllReduce / ncclAllGather (single-rank communicator): PAROPT_SWEEP_RCCL=1 PAROPT_SWEEP_CASES=600 PAROPT_SWEEP_SEED=3333"
  PAROPT_SWEEP_RCCL=1 PAROPT_SWEEP_CASES=600 PAROPT_SWEEP_SEED=3333 timeout 1500 python3 tests/test_gpu_random_sweep.py 2>&1 | grep -v "$F" | cut -c1-1200
) > gpurun_out/r05_sweep_campaigns_more.txt
grep "differ\|ERROR\|^CASE\|^HOST\|^FACADE\|^QN" gpurun_out/r05_sweep_campaigns_more.txt | cut -c1-260
